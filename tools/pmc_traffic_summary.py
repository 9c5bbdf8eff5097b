"""Summarise tools/pmc_traffic.sh output: per-launch HBM bytes of the GEMM kernel family (FETCH_SIZE x2 on gfx950 + WRITE_SIZE;
rocprofv3 reports both in KiB)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd.build import source_digest  # noqa: E402


def load(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if 'gemm_' in r['Kernel_Name'] and '_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value'])
            n += 1
    return tot, n


root = sys.argv[1]
f, nf = load(f'{root}/fetch/p_counter_collection.csv', 'FETCH_SIZE')
w, nw = load(f'{root}/write/p_counter_collection.csv', 'WRITE_SIZE')
out = {'kernel': 'vsx_gemm_f16 (all shapes of one inversion step + one CFG step)', 'launches': nf,
       'lib_digest': source_digest(),
       'fetch_size_kib_raw_per_launch': f / max(nf, 1), 'write_size_kib_per_launch': w / max(nw, 1),
       'hbm_bytes_per_launch': (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
       'note': 'FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE uncalibrated'}
print(json.dumps(out, indent=1))
