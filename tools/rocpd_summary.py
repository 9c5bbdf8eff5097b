"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, as text.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
(rocprofv3 --kernel-trace --stats writes a rocpd database on ROCm 7.2; this is the `--stats` table in plain text.)
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute(f'select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), '
                     f'max(end - start) from kernels group by {name_col} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print(f'# {path}: {sum(r[1] for r in rows)} kernel dispatches, {total / 1e6:.3f} ms of kernel time')
    print(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"share":>6s}  kernel')
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        print(f'{n:7d} {tot / 1e6:10.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:5.1f}%  {short}')
    # idle time between consecutive dispatches (gaps above 1 ms are phase boundaries / host work, listed apart)
    ev = c.execute('select start, end from kernels order by start').fetchall()
    small = big = 0
    nsmall = nbig = 0
    hi = ev[0][1] if ev else 0
    for st, en in ev[1:]:
        if st > hi:
            g = st - hi
            if g < 1_000_000:
                small += g
                nsmall += 1
            else:
                big += g
                nbig += 1
        hi = max(hi, en)
    # distribution of the sub-ms gaps, and which kernel they follow (the dispatch that just finished): a gap is the
    # command processor's dependent-launch latency plus whatever the host had not yet queued
    evn = c.execute(f'select start, end, {name_col} from kernels order by start').fetchall()
    gaps, after = [], {}
    hi2 = evn[0][1] if evn else 0
    prev = evn[0][2] if evn else ''
    for st, en, nm in evn[1:]:
        g = max(0, st - hi2)
        if g < 1_000_000:
            gaps.append(g)
            a = after.setdefault(prev.split('(')[0][-60:], [0, 0])
            a[0] += 1
            a[1] += g
        if en >= hi2:
            hi2, prev = en, nm
    if gaps:
        gs = sorted(gaps)
        print(f'# gaps < 1 ms: median {gs[len(gs) // 2] / 1e3:.2f} us, mean {sum(gs) / len(gs) / 1e3:.2f} us, p90 '
              f'{gs[int(0.9 * len(gs))] / 1e3:.2f} us, p99 {gs[int(0.99 * len(gs))] / 1e3:.2f} us; '
              f'{sum(1 for g in gs if g > 20000)} gaps > 20 us carry {sum(g for g in gs if g > 20000) / 1e6:.2f} ms')
        for k, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f'#   after {k:60s} {n:6d} gaps, {t / 1e6:8.3f} ms, {t / n / 1e3:6.2f} us each')
    if ev:
        span = hi - ev[0][0]
        print(f'# span {span / 1e6:.3f} ms; idle between dispatches: {small / 1e6:.3f} ms in {nsmall} gaps < 1 ms '
              f'({100 * small / span:.1f}% of span), {big / 1e6:.3f} ms in {nbig} gaps >= 1 ms')


if __name__ == '__main__':
    main(sys.argv[1])
