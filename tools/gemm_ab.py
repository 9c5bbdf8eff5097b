"""A/B of the GEMM back ends at the UNet's real shapes: the workgroup-per-tile kernels (option gemm_pp = 0) against the
persistent ping-pong kernel with 256-row tiles (gemm_pp = 2), with 128-row tiles (gemm_pp = 3) and the automatic choice
(gemm_pp = 1, what the product runs).

    python tools/gemm_ab.py [--batch 2] [--rounds 5] > gpurun_out/gemm_ab.txt
    python tools/gemm_ab.py --scheds 0,8                # option bits of pp_sched (gemm_common.h PP_*):
                                                                          # tile kernels vs 256-row persistent tiles per schedule

Variants are interleaved round by round inside ONE process (a cross-process comparison has >3 % noise); the table shows
the median of the per-round times, TFLOP/s of the best variant and the speed-up over the tile kernels.  Shapes: one
UNet forward at B x 16 frames, 64x64 latent (count = launches per forward)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoswap_amd import ops  # noqa: E402

DEV, H16 = 'cuda', torch.float16


def r(*s, scale=1.0):
    return (torch.randn(*s, device=DEV, dtype=torch.float32) * scale).to(H16)


def shapes(B):
    BF = 16 * B
    out = []
    lv = [(64, 320), (32, 640), (16, 1280), (8, 1280)]
    for hw, c in lv:
        M = BF * hw * hw
        n_blk = {64: 5, 32: 5, 16: 5, 8: 0}[hw]          # transformer blocks at this level (spatial == temporal count)
        if n_blk:
            out.append(('plain', f'proj {c}->{c} +res', dict(M=M, N=c, K=c, res=True), 9 * n_blk))
            out.append(('plain', f'qkv {c}->{3 * c}', dict(M=M, N=3 * c, K=c, res=False), 2 * n_blk))
            out.append(('plain', f'qk {c}->{2 * c}', dict(M=M, N=2 * c, K=c, res=False), n_blk))
            out.append(('geglu', f'geglu {c}->{4 * c}', dict(M=M, N=4 * c, K=c), 2 * n_blk))
            out.append(('plain', f'ff2 {4 * c}->{c} +res', dict(M=M, N=c, K=4 * c, res=True), 2 * n_blk))
            # V^T projection of the spatial self-attention (transposed store, LayerNorm folded in): one per transformer block
            out.append(('vt', f'v^T {c}->{c} (+LN)', dict(M=M, N=c, K=c, rows=hw * hw), n_blk))
    convs = [(64, 320, 0, 320, 1, False, 7), (64, 320, 320, 320, 1, False, 2), (64, 640, 320, 320, 1, False, 1),
             (64, 320, 0, 320, 2, False, 1), (32, 640, 0, 640, 1, False, 6), (32, 640, 640, 640, 1, False, 1),
             (32, 320, 0, 640, 1, False, 1), (32, 1280, 640, 640, 1, False, 1), (32, 640, 0, 640, 1, True, 1),
             (16, 1280, 0, 1280, 1, False, 6), (16, 1280, 1280, 1280, 1, False, 2), (16, 640, 0, 1280, 1, False, 1),
             (16, 1280, 0, 1280, 1, True, 1), (8, 1280, 0, 1280, 1, False, 11), (8, 1280, 1280, 1280, 1, False, 3)]
    for hw, c1, c2, co, st, up, n in convs:
        out.append(('conv', f'conv3x3 {hw}x{hw} {c1}+{c2}->{co}' + ('/s2' if st == 2 else '') + (' up' if up else ''),
                    dict(nimg=BF, hw=hw, c1=c1, c2=c2, co=co, stride=st, up=up), n))
    return out


def pack_b(w):
    """piece-major copy of a K-major weight ([N, K] or [N, kh, kw, C]): [N/8][K/64][8 rows][64 halfs], same shape"""
    n = w.shape[0]
    k = w.numel() // n
    assert n % 8 == 0 and k % 64 == 0
    return w.reshape(n // 8, 8, k // 64, 64).permute(0, 2, 1, 3).contiguous().view(w.shape)


def make(kind, a, packed=False):
    """-> (launch closure, algorithmic FLOP); packed: the closure passes the piece-major weight (development library,
    pp_sched + 16)"""
    pk = pack_b if packed else (lambda t: t)
    if kind == 'vt':
        M, N, K = a['M'], a['N'], a['K']
        x, w = r(M, K), pk(r(N, K, scale=K ** -0.5))
        gamma, beta = r(K), r(K)
        return (lambda: ops.linear_vt(ops.DeferredLN(x, gamma, beta, 1e-5), w, None, a['rows'])), 2.0 * M * N * K
    if kind in ('plain', 'geglu'):
        M, N, K = a['M'], a['N'], a['K']
        x = r(M, K)
        if kind == 'geglu':
            w, b = r(2 * N, K, scale=K ** -0.5), r(2 * N)
            w = pk(w)
            return (lambda: ops.linear(x, w, b, geglu=True)), 2.0 * M * 2 * N * K
        w, b = r(N, K, scale=K ** -0.5), r(N)
        w = pk(w)
        res = r(M, N) if a['res'] else None
        return (lambda: ops.linear(x, w, b, residual=res)), 2.0 * M * N * K
    hw = a['hw'] // 2 if a['up'] else a['hw']
    x = r(a['nimg'], hw, hw, a['c1'])
    x2 = r(a['nimg'], hw, hw, a['c2']) if a['c2'] else None
    cin = a['c1'] + a['c2']
    w, b = r(a['co'], 3, 3, cin, scale=(9 * cin) ** -0.5), r(a['co'])
    w = pk(w)
    ho = a['hw'] // a['stride']
    rv = r(a['nimg'] // 16, a['co'])
    return (lambda: ops.conv2d(x, w, b, x2=x2, stride=a['stride'], upsample=a['up'], rowvec=rv,
                               rows_per_vec=16 * ho * ho)), 2.0 * a['nimg'] * ho * ho * a['co'] * 9 * cin


def time_once(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


_LIBS = {}          # --libs: variant name -> typed ctypes handle of another BUILD of the library (lib/libvsx_<name>.so)


def load_build(name):
    """'product' = the library the product loads; anything else = lib/libvsx_<name>.so (python -m videoswap_amd.build --from-git
    <name> <rev>: the kernel sources of another revision), typed like the product's and swapped in under videoswap_amd.ops for the
    launches of that variant.  Same ABI required."""
    import ctypes
    from videoswap_amd import _lib
    if name == 'product':
        return _lib.load()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libvsx_{name}.so'))
    for fn_name, (restype, argtypes) in _lib.PROTOTYPES.items():
        fn = getattr(lib, fn_name)
        fn.restype, fn.argtypes = restype, argtypes
    assert lib.vsx_abi_version() == _lib.VSX_ABI_VERSION, 'the other build must speak the same ABI'
    return lib


def apply(v):
    """variant = (name, gemm_pp[, pp_sched[, tile_tune[, xcd_walk]]]); with --libs the name selects the BUILD"""
    if _LIBS:
        from videoswap_amd import _lib
        _lib._lib = _LIBS[v[0]]
    ops.set_option('gemm_pp', v[1])
    ops.set_option('pp_sched', v[2] if len(v) > 2 else 0)
    ops.set_option('tile_tune', v[3] if len(v) > 3 else 0)
    ops.set_option('xcd_walk', v[4] if len(v) > 4 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--scheds', default='', help='comma-separated pp_sched values: compare option bits instead of tile sizes')
    ap.add_argument('--variants', default='', help='comma-separated name:gemm_pp:pp_sched:tile_tune[:xcd_walk]')
    ap.add_argument('--base', default='', help='name:gemm_pp:pp_sched:tile_tune[:xcd_walk] of the baseline column (default tile:0:0:0)')
    ap.add_argument('--levels', default='', help='comma-separated latent sides (64,32,16,8): only the shapes of these levels')
    ap.add_argument('--auto-scheds', default='', help='comma-separated pp_sched values under the automatic dispatch (gemm_pp = 1)')
    ap.add_argument('--bm', type=int, default=256, choices=(128, 256), help='row tile of the --scheds comparison')
    ap.add_argument('--kinds', default='', help="comma-separated subset of plain,geglu,conv (default: all)")
    ap.add_argument('--libs', default='', help='comma-separated builds, first = baseline column: "r5,product" compares lib/libvsx_r5.so '
                    '(python -m videoswap_amd.build --from-git r5 <rev>) with the product library under the automatic dispatch')
    args = ap.parse_args()
    variants = [('tile', 0, 0), ('pp256', 2, 0), ('pp128', 3, 0), ('auto', 1, 0)]
    if args.scheds:
        def label(n):           # pp_sched bits: 8 = linear tile walk
            n = int(n)
            return f'pp{args.bm}/s{n}'
        variants = [('tile', 0, 0)] + [(label(n), 2 if args.bm == 256 else 3, int(n)) for n in args.scheds.split(',')]
    if args.variants:           # free form: name:gemm_pp:pp_sched:tile_tune (option "tile_tune": gemm.hip)
        variants = [('tile', 0, 0)] + [(v.split(':')[0],) + tuple(int(x) for x in v.split(':')[1:]) for v in args.variants.split(',')]
    if args.base:
        variants[0] = ('tile',) + tuple(int(x) for x in args.base.split(':')[1:])
    if args.auto_scheds:        # the product's own dispatch (gemm_pp = 1) under different pp_sched bits; 'tile' stays the baseline column
        variants = [('tile', 0, 0)] + [(f'auto/s{int(n)}', 1, int(n)) for n in args.auto_scheds.split(',')]
    if args.libs:               # two BUILDS of the library, both under the product's dispatch; the first is the baseline column
        names = args.libs.replace('+', ',').split(',')
        for n in names:
            _LIBS[n] = load_build(n)
        variants = [(n, 1, 0) for n in names]
        print('# builds: ' + ', '.join(f'{n} = {_LIBS[n].vsx_source_digest().decode()[:16]}' for n in names))
    base = variants[0][0]
    print(f'# B={args.batch} T=16 64x64; median of {args.rounds} rounds x {args.reps} launches; times in us')
    print(f'{"shape":44s} {"n":>3s} ' + ' '.join(f'{v[0]:>9s}' for v in variants) + '   best TF/s  speedup  fwd-ms tile -> best')
    tot_old = tot_best = tot_second = 0.0
    level_of = {16 * args.batch * hw * hw: hw for hw in (64, 32, 16, 8)}     # rows of a plain GEMM -> latent side of its level
    for kind, name, a, count in shapes(args.batch):
        if args.kinds and kind not in args.kinds.split(','):
            continue
        if args.levels:
            lvl = a['hw'] if kind == 'conv' else level_of[a['M']]
            if str(lvl) not in args.levels.replace('+', ',').split(','):
                continue
        torch.manual_seed(0)
        fn, flop = make(kind, a)
        fns = {v[0]: fn for v in variants}
        ts = {v[0]: [] for v in variants}
        outs = {}
        for v in variants:                      # warm every variant (first launch sets the LDS attribute)
            apply(v)
            outs[v[0]] = fns[v[0]]()
        torch.cuda.synchronize()
        bad = [k for k, o in outs.items() if not torch.equal(o, outs[base])]
        if bad:         # another K order (pp_sched bit 4) is not bit-identical: report how far it is
            ref = outs[base].float()
            dev = {k: float((outs[k].float() - ref).norm() / ref.norm()) for k in bad}
            print(f'# {name}: variants differing from the tile kernels (rel-L2): ' + ', '.join(f'{k} {v:.2e}' for k, v in dev.items()))
        del outs
        for rnd in range(args.rounds):
            # the order rotates round by round: a variant's time depends on what ran just before it (the later columns of a
            # fixed order came out 1 - 4 % faster on the long convolutions, profiles/r04_gemm_rotation_ab_b*.txt)
            for v in variants[rnd % len(variants):] + variants[:rnd % len(variants)]:
                apply(v)
                ts[v[0]].append(time_once(fns[v[0]], args.reps))
        med = {k: sorted(x)[len(x) // 2] * 1000.0 for k, x in ts.items()}
        best = min(med, key=med.get)
        tot_old += med[base] * count / 1000.0
        tot_best += med[best] * count / 1000.0
        print(f'{name:44s} {count:3d} ' + ' '.join(f'{med[v[0]]:9.1f}' for v in variants) +
              f'   {flop / med[best] / 1e6:7.1f}  {med[base] / med[best]:6.2f}x  {best:9s}'
              f' {med[base] * count / 1000:6.2f} -> {med[best] * count / 1000:6.2f}', flush=True)
        if _LIBS and len(variants) == 2:
            tot_second += med[variants[1][0]] * count / 1000.0
    if _LIBS:
        from videoswap_amd import _lib
        _lib._lib = _LIBS.get('product', _lib._lib)
        if len(variants) == 2:
            print(f'# GEMM time per forward (listed shapes): {variants[0][0]} {tot_old:.2f} ms, {variants[1][0]} {tot_second:.2f} ms')
        return
    apply(('auto', 1))
    print(f'# GEMM time per forward (listed shapes): tile kernels {tot_old:.2f} ms, best-of {tot_best:.2f} ms')


if __name__ == '__main__':
    main()
