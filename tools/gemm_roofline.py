"""Per-shape roofline of the GEMM launches of one UNet forward, from the A/B tables of tools/gemm_ab.py (no GPU needed):

    python tools/gemm_roofline.py profiles/r02_gemm_ab_b2.txt --batch 2

For every shape: algorithmic FLOP, algorithmic HBM bytes (A once + W once + C once (+ residual once)), the attainable
rate min(MFMA peak, FLOP/byte x HBM peak), the measured rate of the variant the product runs (column `auto`) and the
fraction of attainable.  The last lines give the time-weighted totals: what the launch mix could reach at the two
rooflines, and where the measured time goes."""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PEAK_TF, HBM_TBS = 2500.0, 8.0


def shape_table(B):
    """(name, count, flop, bytes) in the order tools/gemm_ab.py prints them"""
    BF = 16 * B
    out = []
    for hw, c, n_blk in ((64, 320, 5), (32, 640, 5), (16, 1280, 5)):
        M = BF * hw * hw
        def lin(name, N, K, res, count, geglu=False):
            nn = 2 * N if geglu else N
            by = 2 * (M * K + nn * K + M * N + (M * N if res else 0))
            out.append((name, count, 2.0 * M * nn * K, by))
        lin(f'proj {c}->{c} +res', c, c, True, 9 * n_blk)
        lin(f'qkv {c}->{3 * c}', 3 * c, c, False, 2 * n_blk)
        lin(f'qk {c}->{2 * c}', 2 * c, c, False, n_blk)
        lin(f'geglu {c}->{4 * c}', 4 * c, c, False, 2 * n_blk, geglu=True)
        lin(f'ff2 {4 * c}->{c} +res', c, 4 * c, True, 2 * n_blk)
    convs = [(64, 320, 0, 320, 1, False, 7), (64, 320, 320, 320, 1, False, 2), (64, 640, 320, 320, 1, False, 1),
             (64, 320, 0, 320, 2, False, 1), (32, 640, 0, 640, 1, False, 6), (32, 640, 640, 640, 1, False, 1),
             (32, 320, 0, 640, 1, False, 1), (32, 1280, 640, 640, 1, False, 1), (32, 640, 0, 640, 1, True, 1),
             (16, 1280, 0, 1280, 1, False, 6), (16, 1280, 1280, 1280, 1, False, 2), (16, 640, 0, 1280, 1, False, 1),
             (16, 1280, 0, 1280, 1, True, 1), (8, 1280, 0, 1280, 1, False, 11), (8, 1280, 1280, 1280, 1, False, 3)]
    for hw, c1, c2, co, st, up, n in convs:
        hin = hw // 2 if up else hw
        ho = hw // st
        cin = c1 + c2
        M = BF * ho * ho
        by = 2 * (BF * hin * hin * cin + co * 9 * cin + M * co)
        name = f'conv3x3 {hw}x{hw} {c1}+{c2}->{co}' + ('/s2' if st == 2 else '') + (' up' if up else '')
        out.append((name, n, 2.0 * M * co * 9 * cin, by))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('table')
    ap.add_argument('--batch', type=int, default=2)
    args = ap.parse_args()
    rows = [l for l in open(args.table) if not l.startswith('#') and re.search(r'\d+\.\d+x', l)]
    shapes = shape_table(args.batch)
    assert len(rows) == len(shapes), (len(rows), len(shapes))
    print(f'# {args.table}: B={args.batch}; attainable = min({PEAK_TF:.0f} TF/s, FLOP/B x {HBM_TBS} TB/s); measured = column "auto"')
    print(f'{"shape":44s} {"n":>3s} {"GFLOP":>8s} {"MB":>7s} {"FLOP/B":>7s} {"attain":>7s} {"meas":>7s} {"frac":>5s} {"bound":>5s}')
    t_meas = t_att = t_mfma = flop_all = by_all = launches = 0.0
    for line, (name, count, flop, by) in zip(rows, shapes):
        assert line.startswith(name), (line[:44], name)
        f = line[44:].split()
        assert int(f[0]) == count
        us = float(f[4])                                  # tile pp256 pp128 auto
        ai = flop / by
        att = min(PEAK_TF, ai * HBM_TBS)
        meas = flop / us / 1e6
        print(f'{name:44s} {count:3d} {flop / 1e9:8.1f} {by / 1e6:7.1f} {ai:7.0f} {att:7.0f} {meas:7.0f} {meas / att:5.2f} '
              f'{"hbm" if att < PEAK_TF else "mfma":>5s}')
        t_meas += us * count
        t_att += flop / att / 1e6 * count
        t_mfma += flop / PEAK_TF / 1e6 * count
        flop_all += flop * count
        by_all += by * count
        launches += count
    print(f'# {int(launches)} launches per forward: {flop_all / 1e12:.2f} TFLOP, {by_all / 1e9:.2f} GB algorithmic '
          f'({by_all / launches / 1e6:.1f} MB per launch, {flop_all / by_all:.0f} FLOP/B)')
    print(f'# time per forward: measured {t_meas / 1e3:.2f} ms = {flop_all / t_meas / 1e6:.0f} TF/s; at the per-shape roofline '
          f'{t_att / 1e3:.2f} ms = {flop_all / t_att / 1e6:.0f} TF/s; at MFMA peak alone {t_mfma / 1e3:.2f} ms')
    print(f'# fraction of the per-shape roofline: {t_att / t_meas:.3f}; of MFMA peak: {t_mfma / t_meas:.3f}')


if __name__ == '__main__':
    main()
