#!/usr/bin/env python
"""Per-SHAPE HBM traffic and duration of vsx_gemm_f16: joins rocprofv3's per-dispatch counters with the launch log the
product writes under VSX_GEMM_LOG (videoswap_amd/ops.py: one line per launch, in launch order).  The i-th dispatch of a
GEMM kernel in the counter CSV is the i-th line of the log (split-K combine kernels are not GEMM kernels and are skipped).

    bash tools/pmc_by_shape.sh            # on the GPU box: two --pmc passes (FETCH_SIZE, WRITE_SIZE) + a kernel trace
    python tools/pmc_by_shape.py gpurun_out/pmc_shape > profiles/rNN_gemm_traffic_by_shape.txt

Columns: launches, average duration, algorithmic bytes per launch (A once + W once + C once + residual once), measured
HBM bytes per launch (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md, + WRITE_SIZE; KiB), their ratio, GB/s, TF/s."""
import csv
import os
import sys
from collections import OrderedDict


def is_gemm(name):
    return 'gemm_' in name and '_kernel' in name


def read_counter(path, counter):
    """-> list of (dispatch order key, kernel name, value, start, end) of the GEMM dispatches, in dispatch order"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter or not is_gemm(r['Kernel_Name']):
                continue
            rows.append((int(r.get('Dispatch_Id', len(rows))), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort(key=lambda t: t[0])
    return rows


def read_trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if is_gemm(r['Kernel_Name']):
                rows.append((int(r['Start_Timestamp']), r['Kernel_Name'],
                             int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    rows.sort(key=lambda t: t[0])
    return rows


def find(root, suffix):
    for d, _, files in os.walk(root):
        for n in files:
            if n.endswith(suffix):
                return os.path.join(d, n)
    return None


def label(f):
    M, N, K, nb, a_mode, ks, stride, ups, C1, C2, H, W, geglu, c_mode, res, rv, bias = f
    if a_mode == 1:
        s = f'conv{ks}x{ks} {H}x{W} {C1}+{C2}->{N}' + ('/s2' if stride == 2 else '') + (' up' if ups else '')
    else:
        s = f'gemm M={M} {K}->{N}' + (' geglu' if geglu else '') + (' vT' if c_mode == 1 else '')
    if nb > 1:
        s += f' x{nb}'
    return s + (' +res' if res else '') + (' +rowvec' if rv else '')


def algorithmic_bytes(f):
    M, N, K, nb, a_mode, ks, stride, ups, C1, C2, H, W, geglu, c_mode, res, rv, bias = f
    cols = N * (2 if geglu else 1)
    if a_mode == 1:
        hin, win = (H // 2, W // 2) if ups else (H, W)
        a = (M // max((H // stride) * (W // stride), 1)) * hin * win * (C1 + C2)      # every input pixel once
    else:
        a = M * K
    return 2 * nb * (a + cols * K + M * N + (M * N if res else 0))


def main(root):
    log = [tuple(int(x) for x in line.split()) for line in open(os.path.join(root, 'gemm_log.txt')) if line.strip()]
    fetch = read_counter(find(os.path.join(root, 'fetch'), 'counter_collection.csv'), 'FETCH_SIZE')
    write = read_counter(find(os.path.join(root, 'write'), 'counter_collection.csv'), 'WRITE_SIZE')
    trace = read_trace(find(os.path.join(root, 'trace'), 'kernel_trace.csv') or find(os.path.join(root, 'fetch'), 'kernel_trace.csv'))
    n = min(len(log), len(fetch), len(write), len(trace))
    print(f'# {len(log)} logged launches, {len(fetch)} / {len(write)} counter rows, {len(trace)} traced dispatches; joined {n}')
    if not (len(log) == len(fetch) == len(write) == len(trace)):
        print('# WARNING: counts differ; the join is by order and may be shifted')
    groups = OrderedDict()
    for i in range(n):
        g = groups.setdefault(log[i], [0, 0.0, 0.0, 0.0, set()])
        g[0] += 1
        g[1] += fetch[i][2] * 2 * 1024
        g[2] += write[i][2] * 1024
        g[3] += trace[i][2] * 1e-9
        g[4].add(trace[i][1].split('<')[0].split('::')[-1] + '<' + trace[i][1].split('<')[1][:14] if '<' in trace[i][1] else trace[i][1])
    print(f'{"shape":52s} {"n":>4s} {"us":>8s} {"alg MB":>8s} {"hbm MB":>8s} {"ratio":>6s} {"rd MB":>8s} {"wr MB":>8s} '
          f'{"GB/s":>7s} {"TF/s":>7s}  kernel')
    tot_alg = tot_hbm = tot_t = 0.0
    rows = []
    for f, (cnt, fb, wb, t, kern) in groups.items():
        M, N, K, nb, *_rest = f
        geglu = f[12]
        alg = algorithmic_bytes(f)
        hbm = (fb + wb) / cnt
        flop = 2.0 * M * N * (2 if geglu else 1) * K * nb
        rows.append((t, f'{label(f):52s} {cnt:4d} {1e6 * t / cnt:8.1f} {alg / 1e6:8.1f} {hbm / 1e6:8.1f} {hbm / alg:6.2f} '
                        f'{fb / cnt / 1e6:8.1f} {wb / cnt / 1e6:8.1f} {hbm / (t / cnt) / 1e9:7.0f} {flop / (t / cnt) / 1e12:7.0f}  '
                        f'{",".join(sorted(kern))}'))
        tot_alg += alg * cnt
        tot_hbm += fb + wb
        tot_t += t
    for _, line in sorted(rows, key=lambda r: -r[0]):
        print(line)
    print(f'# total: algorithmic {tot_alg / 1e9:.2f} GB, measured {tot_hbm / 1e9:.2f} GB = {tot_hbm / tot_alg:.2f}x; '
          f'{tot_hbm / n / 1e6:.1f} MB per launch; GEMM time {tot_t * 1e3:.2f} ms')
    # the aggregate bench.py reports as roofline.traffic (valid for the library build whose digest it carries)
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from videoswap_amd.build import source_digest
    meta = {}
    if os.path.exists(os.path.join(root, 'gemm_log.txt.meta.json')):
        meta = json.load(open(os.path.join(root, 'gemm_log.txt.meta.json')))
    cps = int(meta.get('clips_per_step', 1))
    entry = {'kernel': 'vsx_gemm_f16 (all shapes of one inversion step + one CFG step)', 'launches': n,
             'clips_per_step': cps,
             'lib_digest': source_digest(), 'hbm_bytes_per_launch': tot_hbm / n,
             'algorithmic_bytes_per_launch': tot_alg / n, 'ratio': tot_hbm / tot_alg,
             'note': 'FETCH_SIZE x 2 (gfx950 reports half the bytes of wide coalesced reads) + WRITE_SIZE, separate '
                     '--pmc passes, tools/pmc_by_shape.sh'}
    with open(os.path.join(root, 'gemm_hbm_traffic.json'), 'w') as f:      # one entry per clips-per-step: {"b1": ..., "b4": ...}
        json.dump({'b%d' % cps: entry}, f, indent=1)


def merge(dst, sources):
    """python tools/pmc_by_shape.py --merge profiles/gemm_hbm_traffic.json gpurun_out/<tag>_gemm_hbm_traffic_b1.json ... :
    the entries of the source files replace the entries of the same batch in dst (bench.py reads dst, keyed by batch)."""
    import json
    cur = {}
    if os.path.exists(dst):
        cur = json.load(open(dst))
        if 'hbm_bytes_per_launch' in cur:            # round-5 layout: one unkeyed entry
            cur = {'b%d' % int(cur.get('clips_per_step', 1)): cur}
    for src in sources:
        new = json.load(open(src))
        if 'hbm_bytes_per_launch' in new:
            new = {'b%d' % int(new.get('clips_per_step', 1)): new}
        cur.update(new)
    with open(dst, 'w') as f:
        json.dump(dict(sorted(cur.items())), f, indent=1)
    print({k: (v['lib_digest'][:12], round(v['ratio'], 3)) for k, v in cur.items()})


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--merge':
        merge(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_shape')
