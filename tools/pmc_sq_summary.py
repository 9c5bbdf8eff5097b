#!/usr/bin/env python
"""Per-kernel sums of the SQ / GRBM counters collected by tools/pmc_sq.sh (rocprofv3 counter_collection CSVs).

    python tools/pmc_sq_summary.py gpurun_out/pmc_sq > profiles/rNN_pmc_sq.txt

Reported per kernel name (template instantiation): dispatches, and for every counter its sum over the dispatches;
derived: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-normalised CU cycles) as the guide defines it
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles = 32 x N_mfma(32x32x16) per SIMD; GRBM_GUI_ACTIVE = wall cycles of the
dispatch, SUMMED over the 8 XCDs by rocprofv3), i.e. mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs);
cross-check of the XCD factor: the conv kernel reads 0.52 at 1.2 PF/s and ~1.95 GHz (profiles/r02_pmc_sq.txt)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name)
    return name.replace('void ', '')[:80]


def main(root):
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    for path in sorted(glob.glob(os.path.join(root, '*', '**', '*counter_collection.csv'), recursive=True)):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                k = short(row.get('Kernel_Name') or row.get('kernel_name') or '?')
                c = row.get('Counter_Name') or row.get('counter_name')
                v = float(row.get('Counter_Value') or row.get('counter_value') or 0)
                sums[k][c] += v
                counts[k][c] += 1
    if not sums:
        print('no counter_collection.csv found under', root)
        return
    allc = sorted({c for k in sums for c in sums[k]})
    print('# counters:', ' '.join(allc))
    order = sorted(sums, key=lambda k: -sums[k].get('GRBM_GUI_ACTIVE', sums[k].get('SQ_WAVE_CYCLES', 0)))
    for k in order:
        s = sums[k]
        n = max(counts[k].values())
        line = f'{k:80s} n={n:5d}'
        gui = s.get('GRBM_GUI_ACTIVE', 0.0)
        if gui and 'SQ_VALU_MFMA_BUSY_CYCLES' in s:
            line += f'  mfma_util={s["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 128):6.3f}'
        wc = s.get('SQ_WAVE_CYCLES', 0.0)
        if wc:
            for c, tag in (('SQ_WAIT_INST_ANY', 'issue_stall'), ('SQ_WAIT_ANY', 'parked'),
                           ('SQ_ACTIVE_INST_ANY', 'active'), ('SQ_WAIT_INST_LDS', 'lds_issue_stall')):
                if c in s:
                    line += f'  {tag}={s[c] / wc:5.3f}'
        if s.get('SQ_LDS_IDX_ACTIVE'):
            line += f'  lds_conflict={s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"]:5.3f}'
        print(line)
        print('    ' + '  '.join(f'{c}={s[c]:.4g}' for c in allc if c in s))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_sq')
