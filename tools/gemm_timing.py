"""Where does a GEMM workgroup spend its main loop?  Builds a -DVSX_GEMM_TIMING copy of the library (per-wave cycle
totals of: MFMA+fragment segment / DMA-landing wait / barrier wait / DMA issue), runs one forced tile shape and prints
the per-slab averages.

    python tools/gemm_timing.py build                     # here (cross-compile)
    VSX_TUNE_TILE=1 python tools/gemm_timing.py run M N K   # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'videoswap_amd', 'lib', 'libvsx_timing.so')


def build():
    from videoswap_amd import build as b
    objs = []
    for src in b.SOURCES:
        obj = os.path.join(b.OBJDIR, 'timing_' + src + '.o')
        subprocess.check_call([b.HIPCC] + b.FLAGS + ['-DVSX_GEMM_TIMING', '-c', os.path.join(b.CSRC, src), '-o', obj])
        objs.append(obj)
    subprocess.check_call([b.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    print('built', LIB)


def run(M, N, K, iters=5):
    import torch
    from videoswap_amd import _lib
    _lib.LIB_PATH = LIB
    lib = _lib.load()
    tile = int(os.environ.get('VSX_TUNE_TILE', '1'))
    BM, BN, NW = {1: (128, 320, 8), 2: (128, 160, 4), 3: (256, 320, 16), 4: (128, 128, 4), 5: (64, 128, 4), 6: (64, 64, 4)}[tile & 15]
    x = torch.randn(M, K, device='cuda', dtype=torch.float16)
    w = torch.randn(N, K, device='cuda', dtype=torch.float16) * 0.02
    out = torch.empty(M, N, device='cuda', dtype=torch.float16)
    nblk = ((M + BM - 1) // BM) * ((N + BN - 1) // BN)
    ws = torch.zeros(nblk * NW * 8, dtype=torch.int64, device='cuda')
    d = _lib.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch0 = d.batch1 = 1
    d.A = x.data_ptr(); d.lda = K
    d.B = w.data_ptr(); d.ldb = K
    d.C = out.data_ptr(); d.ldc = N
    d.alpha = 1.0
    if os.environ.get('VSX_TIMING_RES'):
        res = torch.randn(M, N, device='cuda', dtype=torch.float16)
        bias = torch.randn(N, device='cuda', dtype=torch.float16)
        d.residual = res.data_ptr(); d.ldr = N
        d.bias = bias.data_ptr()
    d.workspace = ws.data_ptr(); d.workspace_bytes = ws.numel() * 8
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(iters):
        _lib.check(lib.vsx_gemm_f16(ctypes.byref(d), ctypes.c_void_p(s)), 'gemm')
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(lib.vsx_gemm_f16(ctypes.byref(d), ctypes.c_void_p(s)), 'gemm')
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 1e3
    full = ws.view(nblk, NW, 8)
    t = full[:, :, :4].double()
    nslab = (K + 63) // 64
    per = t.mean(dim=(0, 1)) / max(nslab - 1, 1)
    tot = per.sum().item()
    names = ['MFMA + fragment reads', 'wait: DMA landed', 'wait: barrier', 'DMA issue (+ frag/MFMA tail)']
    print(f'tile {BM}x{BN} ({NW} waves)  M={M} N={N} K={K}: {nblk} workgroups, {us:.1f} us, '
          f'{2.0 * M * N * K / us / 1e6:.0f} TFLOP/s; per-slab cycles (mean over waves): total {tot:.0f}')
    for n, v in zip(names, per.tolist()):
        print(f'    {n:32s} {v:8.0f} cyc  {100 * v / tot:5.1f}%')
    wv = t.mean(dim=0) / max(nslab - 1, 1)
    print('    per-wave totals:', [f'{v:.0f}' for v in wv.sum(dim=1).tolist()])
    phases(full, us)


def phases(full, us):
    """Absolute stamps (entries 4-7 of a wave's record: kernel entry, first slab multiplied, last slab multiplied, stores
    acknowledged): where the launch's wall time goes besides the K loop."""
    st = full[:, :, 4:8].double()
    t0 = st[:, :, 0].min().item()
    span = st[:, :, 3].max().item() - t0
    rel = st - t0
    b, l0, l1, e = (rel[:, :, i] for i in range(4))
    print(f'    launch span {span:.0f} cycles = {us:.1f} us by the events -> ~{span / us / 1e3:.2f} GHz if the span were the whole launch')
    print(f'    entry (first ... last wave)        {b.min().item():8.0f} ... {b.max().item():8.0f}')
    print(f'    prologue: entry -> first slab     mean {(l0 - b).mean().item():8.0f}  max {(l0 - b).max().item():8.0f}')
    print(f'    K loop                            mean {(l1 - l0).mean().item():8.0f}  max {(l1 - l0).max().item():8.0f}')
    print(f'    epilogue (stores acknowledged)    mean {(e - l1).mean().item():8.0f}  max {(e - l1).max().item():8.0f}')
    print(f'    exit (first ... last wave)         {e.min().item():8.0f} ... {e.max().item():8.0f}')


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'run':
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))


def run_attn(nb, heads, nq, nk, d):
    import torch
    from videoswap_amd import _lib, ops
    _lib.LIB_PATH = LIB
    lib = _lib.load()
    C = heads * d
    q = torch.randn(nb, nq, C, device='cuda', dtype=torch.float16)
    k = torch.randn(nb, nk, C, device='cuda', dtype=torch.float16)
    v = torch.randn(nb, nk, C, device='cuda', dtype=torch.float16)
    nblk = ((nq + 127) // 128) * heads * nb
    dbg = torch.zeros(nblk * 4 * 6, dtype=torch.int64, device='cuda')
    lib.vsx_attn_debug_buffer.restype = ctypes.c_int
    lib.vsx_attn_debug_buffer.argtypes = [ctypes.c_void_p]
    assert lib.vsx_attn_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
    wv = torch.randn(C, C, device='cuda', dtype=torch.float16) * 0.05
    vt = ops.linear_vt(v.view(-1, C), wv, None, nk)      # [nb, C, ld] V^T
    for _ in range(3):
        out = ops.attention(q, k, vt, heads, d ** -0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = ops.attention(q, k, vt, heads, d ** -0.5)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 1e3
    t = dbg.view(nblk, 4, 6).double()
    nit = (nk + 63) // 64
    per = t.mean(dim=(0, 1)) / nit
    tot = per.sum().item()
    names = ['wait: K/V tile landed', 'wait: barrier', 'DMA issue', 'S^T = K Q^T (MFMA)', 'online softmax (VALU)', 'O^T += V^T P^T (MFMA)']
    print(f'attention nb={nb} heads={heads} nq={nq} nk={nk} d={d}: {us:.1f} us, {4.0 * nb * nq * nk * C / us / 1e6:.0f} TFLOP/s; '
          f'cycles per 64-key tile (mean over waves): {tot:.0f}')
    for n, x in zip(names, per.tolist()):
        print(f'    {n:28s} {x:8.0f} cyc  {100 * x / tot:5.1f}%')


def run_pp():
    """Persistent GEMM: cycles a workgroup spends in its K loops / its epilogues / the barrier behind them, per tile, for
    the shapes that carry the UNet's GEMM time (B = 2, T = 16, 64x64 and 32x32 levels)."""
    import torch
    from videoswap_amd import _lib, ops
    _lib.LIB_PATH = LIB
    lib = _lib.load()
    lib.vsx_pp_debug_buffer.restype = ctypes.c_int
    lib.vsx_pp_debug_buffer.argtypes = [ctypes.c_void_p]
    dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device='cuda')
    assert lib.vsx_pp_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
    dev, h16 = 'cuda', torch.float16

    def r(*s, scale=1.0):
        return (torch.randn(*s, device=dev) * scale).to(h16)
    cases = []
    for M, K, N, kind in ((131072, 320, 1280, 'geglu'), (131072, 320, 320, 'res'), (131072, 320, 960, 'plain'),
                          (131072, 1280, 320, 'res'), (32768, 640, 2560, 'geglu'), (32768, 640, 640, 'res'),
                          (32768, 2560, 640, 'res'), (8192, 1280, 5120, 'geglu')):
        x = r(M, K)
        if kind == 'geglu':
            w, b = r(2 * N, K, scale=K ** -0.5), r(2 * N)
            cases.append((f'geglu M={M} {K}->{N}', (lambda x=x, w=w, b=b: ops.linear(x, w, b, geglu=True)), 2.0 * M * 2 * N * K))
        else:
            w, b = r(N, K, scale=K ** -0.5), r(N)
            res = r(M, N) if kind == 'res' else None
            cases.append((f'gemm M={M} {K}->{N}' + (' +res' if res is not None else ''),
                          (lambda x=x, w=w, b=b, res=res: ops.linear(x, w, b, residual=res)), 2.0 * M * N * K))
    xc = r(32, 64, 64, 320)
    wc, bc = r(320, 3, 3, 320, scale=(9 * 320) ** -0.5), r(320)
    cases.append(('conv3x3 64x64 320->320', (lambda: ops.conv2d(xc, wc, bc)), 2.0 * 32 * 4096 * 320 * 9 * 320))
    print('shape                               us    TF/s   tiles/WG   cycles per tile: K loop   epilogue   barrier    epilogue share')
    for name, fn, flop in cases:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dbg.zero_()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b_.record(); b_.synchronize()
        us = a.elapsed_time(b_) * 1e3
        t = dbg.view(256, 8, 4).double()
        live = t[:, :, 3] > 0
        if not bool(live.any()):
            print(f'{name:32s} {us:7.1f}  (not on the persistent kernel)')
            continue
        tiles = t[:, :, 3][live].mean().item()
        per = [(t[:, :, i][live] / t[:, :, 3][live]).mean().item() for i in range(3)]
        print(f'{name:32s} {us:7.1f} {flop / us / 1e6:7.0f} {tiles:9.1f} {per[0]:24.0f} {per[1]:10.0f} {per[2]:9.0f} '
              f'{per[1] / sum(per):16.2f}')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'pp':
    run_pp()
if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'attn':
    run_attn(*[int(x) for x in sys.argv[2:7]])
