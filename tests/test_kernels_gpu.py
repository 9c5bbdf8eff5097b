"""GPU parity tests of every libvsx kernel against a plain PyTorch fp32 reference of the same op.

Inputs are fp16 (exactly representable in the fp32 reference), so the only differences are the
accumulation order and the final fp16 rounding: tolerances are a few fp16 ulps of the output scale.
All calls go through the C ABI (videoswap_amd.ops -> ctypes -> libvsx.so).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def ops():
    from videoswap_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16).to(DEV)


def rel_err(out, ref, l2_tol=1.5e-3, row_tol=6e-3):
    """Returns max|d| / max|ref| (the caller's threshold) and ASSERTS two metrics the max-norm cannot see:
    the rel-L2 error of the whole tensor (a wrong low-magnitude region) and the worst per-row rel-L2 error
    (one bad tile edge / one bad row of an otherwise fine tensor; rows = the last axis, rows whose reference norm is
    below 10 % of the mean row norm are measured against that floor).  An fp16-rounded exact result sits at
    rel-L2 ~3e-4."""
    out = out.float()
    ref = ref.float()
    assert out.shape == ref.shape, f'shape {tuple(out.shape)} vs {tuple(ref.shape)}'
    assert torch.isfinite(out).all(), 'non-finite values in kernel output'
    d = (out - ref).double()
    r = ref.double()
    l2 = (d.norm() / r.norm().clamp_min(1e-12)).item()
    assert l2 < l2_tol, f'rel-L2 {l2:.3e} >= {l2_tol:.1e}'
    if ref.dim() >= 2 and ref.shape[-1] >= 8:
        dr, rr = d.reshape(-1, d.shape[-1]).norm(dim=1), r.reshape(-1, r.shape[-1]).norm(dim=1)
        row = (dr / rr.clamp_min(0.1 * rr.mean().clamp_min(1e-12))).max().item()
        assert row < row_tol, f'worst row rel-L2 {row:.3e} >= {row_tol:.1e} (tensor rel-L2 {l2:.3e})'
    return (d.abs().max() / r.abs().max().clamp_min(1e-6)).item()


# --------------------------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(256, 256, 128), (300, 200, 72), (1024, 320, 320), (2, 1280, 320),
                                   (8192, 1280, 640), (77, 64, 768), (131, 40, 8)])
def test_linear(M, N, K):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    out = ops().linear(x, w, b, residual=r)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    assert rel_err(out, ref) < 2e-3
    out2 = ops().linear(x, w)
    assert rel_err(out2, x.float() @ w.float().t()) < 2e-3


@pytest.mark.parametrize('M,N,K', [(512, 256, 128), (100, 72, 64), (8192, 1280, 320)])
def test_linear_geglu(M, N, K):
    x, w, b = rnd(M, K, seed=5), rnd(2 * N, K, seed=6, scale=K ** -0.5), rnd(2 * N, seed=7)
    out = ops().linear(x, w, b, geglu=True)
    y = x.float() @ w.float().t() + b.float()
    ref = y[:, :N] * F.gelu(y[:, N:])
    assert out.shape == (M, N)
    assert rel_err(out, ref) < 2e-3


@pytest.mark.parametrize('rows,nimg,N,K', [(64, 6, 64, 128), (77, 2, 320, 768), (256, 3, 40, 64)])
def test_linear_vt(rows, nimg, N, K):
    x, w, b = rnd(nimg * rows, K, seed=8), rnd(N, K, seed=9, scale=K ** -0.5), rnd(N, seed=10)
    vt = ops().linear_vt(x, w, b, rows)
    ref = (x.float() @ w.float().t() + b.float()).view(nimg, rows, N).transpose(1, 2)
    assert vt.shape[2] % 8 == 0
    assert rel_err(vt[:, :, :rows], ref) < 2e-3
    if vt.shape[2] != rows:
        assert (vt[:, :, rows:] == 0).all()


def conv_ref(x, w, b, stride, x2=None, upsample=False):
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    xin = xin.permute(0, 3, 1, 2)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ks = w.shape[1]
    wt = w.float().permute(0, 3, 1, 2)
    y = F.conv2d(xin, wt, b.float() if b is not None else None, stride=stride, padding=ks // 2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize('nimg,H,W,C1,C2,Cout,ks,stride,ups', [
    (2, 16, 16, 64, 0, 64, 3, 1, False),
    (3, 12, 20, 8, 0, 32, 3, 1, False),      # conv_in-like (Cin padded to 8), non-square
    (2, 16, 16, 64, 0, 128, 3, 2, False),    # Downsample3D
    (2, 8, 12, 64, 0, 64, 3, 1, True),       # Upsample3D (nearest 2x folded into the loader)
    (2, 16, 16, 128, 64, 64, 3, 1, False),   # skip concat folded into the loader
    (2, 16, 16, 128, 64, 96, 1, 1, False),   # 1x1 shortcut on a concat
    (4, 64, 64, 320, 0, 320, 3, 1, False),   # big-tile path
    (2, 14, 24, 320, 0, 4, 3, 1, False),     # conv_out-like (N = 4)
    # M >= 61 440 at N = 320 / 640: the 256x320 tile that carries the big-M convolutions of the benchmarked model
    (16, 64, 64, 320, 0, 320, 3, 1, False),      # M = 65 536: ResnetBlock3D conv at the 64x64 level (B*F = 16)
    (16, 64, 64, 320, 320, 320, 3, 1, False),    # two-source (skip concat) conv1 of the last up block
    (16, 64, 64, 320, 0, 320, 1, 1, False),      # 1x1 on the big tile (K = 320: five slabs)
    (64, 64, 64, 320, 0, 640, 3, 2, False),      # Downsample3D-shaped stride 2: M = 65 536, N = 640
    (16, 32, 32, 640, 0, 320, 3, 1, True),       # Upsample3D: half-resolution source, M = 65 536
    (20, 56, 96, 320, 0, 320, 3, 1, False),      # 448x768 frames (56x96 latent): M = 107 520, ragged last M tile
    (64, 32, 32, 640, 0, 640, 3, 1, False),      # M = 65 536, N = 640 (two column tiles)
])
def test_conv2d(nimg, H, W, C1, C2, Cout, ks, stride, ups):
    x = rnd(nimg, H, W, C1, seed=11)
    x2 = rnd(nimg, H, W, C2, seed=12) if C2 else None
    K = ks * ks * (C1 + C2)
    w = rnd(Cout, ks, ks, C1 + C2, seed=13, scale=K ** -0.5)
    b = rnd(Cout, seed=14)
    out = ops().conv2d(x, w, b, x2=x2, stride=stride, upsample=ups)
    ref = conv_ref(x, w, b, stride, x2, ups)
    assert rel_err(out, ref) < 2e-3


@pytest.mark.parametrize('M,N,K', [(2048, 1280, 2560), (300, 640, 4096), (1024, 320, 5120)])
def test_linear_split_k(M, N, K):
    """small-M / long-K problems are sliced along K (fp32 partials + deterministic combine kernel)"""
    x, w, b, r = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5), rnd(N, seed=63), rnd(M, N, seed=64)
    out = ops().linear(x, w, b, residual=r)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    assert rel_err(out, ref) < 2e-3
    assert torch.equal(out, ops().linear(x, w, b, residual=r))       # deterministic


@pytest.mark.parametrize('nimg,H,W,C1,C2,Cout,stride,ups', [
    (2, 8, 8, 256, 0, 640, 1, False),        # 72 slabs -> 3 slices
    (4, 8, 8, 128, 192, 320, 1, False),      # two sources: the tap counters start mid-way in a slice
    (2, 16, 16, 320, 0, 320, 2, False),      # stride 2
    (2, 4, 4, 640, 0, 640, 1, True),         # nearest-2x upsample folded in
])
def test_conv2d_split_k(nimg, H, W, C1, C2, Cout, stride, ups):
    x = rnd(nimg, H, W, C1, seed=65)
    x2 = rnd(nimg, H, W, C2, seed=66) if C2 else None
    K = 9 * (C1 + C2)
    w, b = rnd(Cout, 3, 3, C1 + C2, seed=67, scale=K ** -0.5), rnd(Cout, seed=68)
    Ho = (2 * H if ups else H) // stride
    Wo = (2 * W if ups else W) // stride
    rowvec, res = rnd(nimg, Cout, seed=69), rnd(nimg, Ho, Wo, Cout, seed=70)
    out = ops().conv2d(x, w, b, x2=x2, stride=stride, upsample=ups, rowvec=rowvec, rows_per_vec=Ho * Wo, residual=res)
    ref = conv_ref(x, w, b, stride, x2, ups) + rowvec.float()[:, None, None, :] + res.float()
    assert rel_err(out, ref) < 2e-3


def test_conv2d_big_tile_rowvec_residual():
    """rowvec (time embedding, one row per batch item) + residual epilogue on the 256x320 tile (M = 65 536)."""
    nimg, H, W, C, Cout = 16, 64, 64, 320, 320
    x, w, b = rnd(nimg, H, W, C, seed=71), rnd(Cout, 3, 3, C, seed=72, scale=(9 * C) ** -0.5), rnd(Cout, seed=73)
    rowvec = rnd(2, Cout, seed=74)
    res = rnd(nimg, H, W, Cout, seed=75)
    out = ops().conv2d(x, w, b, rowvec=rowvec, rows_per_vec=8 * H * W, residual=res)
    ref = conv_ref(x, w, b, 1) + rowvec.float().repeat_interleave(8, 0)[:, None, None, :] + res.float()
    assert rel_err(out, ref) < 2e-3
    assert torch.equal(out, ops().conv2d(x, w, b, rowvec=rowvec, rows_per_vec=8 * H * W, residual=res))


@pytest.mark.parametrize('M,N,K,res', [(65536, 320, 320, True), (65536, 320, 1280, True), (65536, 960, 320, False),
                                       (61440, 640, 640, False), (131072, 320, 320, False)])
def test_linear_big_m(M, N, K, res):
    """the projections of the 64x64 level at the benchmarked width (256x320 / 128x160 tiles, XCD-remapped grids)"""
    x, w, b = rnd(M, K, seed=76), rnd(N, K, seed=77, scale=K ** -0.5), rnd(N, seed=78)
    r = rnd(M, N, seed=79) if res else None
    out = ops().linear(x, w, b, residual=r)
    ref = x.float() @ w.float().t() + b.float()
    if res:
        ref = ref + r.float()
    assert rel_err(out, ref) < 2e-3


def test_linear_geglu_big_m():
    M, N, K = 65536, 1280, 320
    x, w, b = rnd(M, K, seed=80), rnd(2 * N, K, seed=81, scale=K ** -0.5), rnd(2 * N, seed=82)
    out = ops().linear(x, w, b, geglu=True)
    y = x.float() @ w.float().t() + b.float()
    assert rel_err(out, y[:, :N] * F.gelu(y[:, N:])) < 2e-3


# --------------------------------------------------------------------------------------------
# persistent ping-pong kernel (gemm_pp.hip) against the reference AND bit-for-bit against the workgroup-per-tile
# kernels (same MFMA, same k order per output element, same epilogue arithmetic)
# --------------------------------------------------------------------------------------------
def both_gemm_paths(fn):
    """fn() under option gemm_pp = 0 (tile kernels only) and 2 (persistent kernel wherever eligible)."""
    o = ops()
    try:
        o.set_option('gemm_pp', 0)
        a = fn()
        o.set_option('gemm_pp', 2)
        b = fn()
    finally:
        o.set_option('gemm_pp', 1)
        o.set_option('pp_sched', 0)
    return a, b


def _sched(sched):
    """pp_sched of the persistent kernel under test (8 = linear tile walk instead of the 2-D one): the case's own, or
    VSX_TEST_PP_SCHED for every case"""
    import os
    return int(os.environ.get('VSX_TEST_PP_SCHED', sched))


@pytest.mark.parametrize('M,N,K,res,sched', [
    (65536, 320, 320, True, 0), (65536, 320, 1280, True, 8), (65536, 960, 320, False, 8), (61440, 640, 640, False, 0),
    (131072, 320, 320, True, 0), (8192, 1280, 1280, True, 0), (8192, 3840, 1280, False, 0), (32768, 640, 2560, True, 0),
    (65500, 320, 328, True, 0),      # ragged last M tile, K tail (328 = 5 slabs + 8)
    (5000, 640, 64, False, 0),       # one slab per tile: every stream element changes tile
    (4096, 640, 640, True, 0),       # 128x320 tiles
    (2048, 1280, 5120, True, 0),     # 128x320 tiles, long K (split-K on the tile kernels)
])
def test_persistent_linear(M, N, K, res, sched):
    x, w, b = rnd(M, K, seed=90), rnd(N, K, seed=91, scale=K ** -0.5), rnd(N, seed=92)
    r = rnd(M, N, seed=93) if res else None
    ops().set_option('pp_sched', _sched(sched))
    old, new = both_gemm_paths(lambda: ops().linear(x, w, b, residual=r))
    ref = x.float() @ w.float().t() + b.float()
    if res:
        ref = ref + r.float()
    assert rel_err(new, ref) < 2e-3
    if K < 2560:                     # (split-K on the tile-kernel side sums in a different order)
        assert torch.equal(old, new), 'persistent and tile kernels must agree bit for bit'


@pytest.mark.parametrize('tune', [1, 2, 2 + 16])
def test_tile_kernels_late_residual_prefetch_at_every_loop_length(tune):
    """ADVICE r5 (gemm.hip, RES_LATE): the NRES residual loads of the workgroup-per-tile kernels go out behind the last piece of
    the last slab, and every slab wait after that point counts `vmcnt(later * G + NRES)` — correct only if every wave issues exactly
    NRES loads and the issue iteration is kt_r = max(nloc - 1 - PREFETCH, 0).  The host model (tools/cpu_check, case 22) checks the
    counts with late-landing DMA; this is the same on hardware, for every K-loop length around the ring depth — nloc = 1 ... PREFETCH
    + 4 slabs, i.e. including nloc = PREFETCH + 2 where the last slab's pieces sit right inside the allowed-in-flight window — on
    the 8-wave 128 x 320 tile (tile_tune 1: the late waves issue their pieces one k-step later), the 4-wave 128 x 160 tile with two
    ring slots (2) and with four (2 + 16), with ragged M (rows past M are clamped, not skipped: NRES loads per wave).  A slab read
    before it landed would show against the fp32 reference AND against the early placement (pp_sched bit 64: residual in front of
    the loop, waits without the exact count), which must agree bit for bit."""
    o = ops()
    prefetch = 3 if tune & 16 else 1
    try:
        o.set_option('gemm_pp', 0)
        o.set_option('tile_tune', tune)
        for nloc in range(1, prefetch + 5):
            K = 64 * nloc
            for M in (4096, 4000):
                x, w, b = rnd(M, K, seed=700 + nloc), rnd(1280, K, seed=701 + nloc, scale=K ** -0.5), rnd(1280, seed=702)
                r = rnd(M, 1280, seed=703 + nloc)
                ref = x.float() @ w.float().t() + b.float() + r.float()
                o.set_option('pp_sched', 0)
                late = [o.linear(x, w, b, residual=r) for _ in range(3)]
                o.set_option('pp_sched', 64)
                early = o.linear(x, w, b, residual=r)
                assert rel_err(late[0], ref) < 2e-3, (tune, nloc, M)
                assert all(torch.equal(t, early) for t in late), (tune, nloc, M)
    finally:
        o.set_option('gemm_pp', 1)
        o.set_option('tile_tune', 0)
        o.set_option('pp_sched', 0)


@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('rows,nimg,N,K', [(4096, 16, 320, 320), (1024, 64, 640, 640), (256, 256, 1280, 1280), (4096, 128, 320, 320)])
def test_persistent_transposed_store(rows, nimg, N, K, ln):
    """The V^T projections of the 64x64 / 32x32 / 16x16 levels on the persistent kernel (round 5: epilogue_vt, full 128-byte lines of
    V^T per wave instead of 8-byte scatters): bit for bit like the tile kernels' transposed store, plain and with the folded LayerNorm
    (what the self-attentions of the model launch); 128 images x 4096 rows = the UNet batch 8 of four clips per step."""
    o = ops()
    x, gamma, beta = _ln_inputs(nimg * rows, K, seed=150)
    w, b = rnd(N, K, seed=151, scale=K ** -0.5), rnd(N, seed=152)
    o.set_option('pp_sched', 0)
    src = (lambda: o.DeferredLN(x, gamma, beta, 1e-5)) if ln else (lambda: x)
    old, new = both_gemm_paths(lambda: o.linear_vt(src(), w, b, rows))
    xin = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5) if ln else x.float()
    ref = (xin @ w.float().t() + b.float()).view(nimg, rows, N).transpose(1, 2)
    assert rel_err(new[:, :, :rows], ref) < 2e-3
    assert torch.equal(old, new), 'persistent and tile kernels must agree bit for bit'


@pytest.mark.parametrize('M,N,K', [(65536, 1280, 320), (32768, 2560, 640), (8192, 5120, 1280), (20000, 160, 320)])
def test_persistent_geglu(M, N, K):
    x, w, b = rnd(M, K, seed=94), rnd(2 * N, K, seed=95, scale=K ** -0.5), rnd(2 * N, seed=96)
    ops().set_option('pp_sched', _sched(0))
    old, new = both_gemm_paths(lambda: ops().linear(x, w, b, geglu=True))
    y = x.float() @ w.float().t() + b.float()
    assert rel_err(new, y[:, :N] * F.gelu(y[:, N:])) < 2e-3
    assert torch.equal(old, new)


@pytest.mark.parametrize('addend', ['rowvec', 'residual', 'none'])
@pytest.mark.parametrize('nimg,H,W,C1,C2,Cout,ks,stride,ups,sched', [
    (16, 64, 64, 320, 0, 320, 3, 1, False, 0),
    (16, 64, 64, 320, 320, 320, 3, 1, False, 8),   # two sources
    (16, 64, 64, 640, 320, 320, 1, 1, False, 0),   # 1x1 shortcut on a concat
    (64, 64, 64, 320, 0, 640, 3, 2, False, 8),     # stride 2
    (16, 32, 32, 640, 0, 320, 3, 1, True, 0),      # nearest-2x upsample folded in
    (20, 56, 96, 320, 0, 320, 3, 1, False, 0),     # ragged M (107 520 rows), Wo = 96
    (32, 16, 16, 1280, 0, 1280, 3, 1, False, 0),   # M = 8192: 128x320 tiles
    (3, 28, 48, 128, 64, 640, 3, 1, False, 0),     # M = 4032 (not a multiple of 128), W = 48: 128x320 tiles
    (100, 6, 8, 128, 0, 320, 3, 1, False, 0),      # 48 rows per image: 32-row blocks that meet two row vectors
    (16, 32, 32, 640, 0, 640, 3, 1, False, 0),     # W = 32: two image rows per wave (shared A slab, see the test body)
    (18, 32, 32, 640, 320, 640, 3, 1, False, 8),   # ... two sources, ragged last tile row (72 tiles of 256)
    (5, 32, 32, 128, 0, 320, 3, 1, False, 0),      # ... 128x320 tiles (M = 5120)
])
def test_persistent_conv(nimg, H, W, C1, C2, Cout, ks, stride, ups, sched, addend):
    """The persistent kernel carries ONE addend through its epilogue ring (time-embedding row vector on a ResNet's first
    convolution, residual on its second); both at once go to the tile kernels (covered by test_conv_*)."""
    x = rnd(nimg, H, W, C1, seed=97)
    x2 = rnd(nimg, H, W, C2, seed=98) if C2 else None
    Kc = ks * ks * (C1 + C2)
    w, b = rnd(Cout, ks, ks, C1 + C2, seed=99, scale=Kc ** -0.5), rnd(Cout, seed=100)
    Ho = (2 * H if ups else H) // stride
    Wo = (2 * W if ups else W) // stride
    rowvec = rnd(nimg, Cout, seed=101) if addend == 'rowvec' else None
    res = rnd(nimg, Ho, Wo, Cout, seed=102) if addend == 'residual' else None
    def run():      # (the nine-tap form of a nearest-2x convolution: its sub-pixel form has a test of its own below)
        ops().CONV_SUBPIXEL = False
        try:
            return ops().conv2d(x, w, b, x2=x2, stride=stride, upsample=ups, rowvec=rowvec,
                                rows_per_vec=Ho * Wo if rowvec is not None else 0, residual=res)
        finally:
            ops().CONV_SUBPIXEL = True
    # pp_sched bit 4: the tap-major K order, which sums like the tile kernels (bit for bit); the default order (the taps of a
    # 64-channel slab back to back: one fabric read of the input window instead of one per tap) is another fp32 summation order
    ops().set_option('pp_sched', _sched(sched) | 4)
    old, same_order = both_gemm_paths(run)
    ops().set_option('pp_sched', _sched(sched))
    _, new = both_gemm_paths(run)
    # stride-1 3x3 convolutions with image rows of 32 / 64 / ... pixels stream ONE A slab per filter row and read it at three
    # row offsets (gemm_pp.hip "SHARED A SLAB"); pp_sched bit 16 keeps a private slab per tap: same products, same order
    # bit 32: every CU issues the pieces of a slab in the same order (default: from a per-CU starting point)
    ops().set_option('pp_sched', _sched(sched) | 16 | 32)
    _, private_a = both_gemm_paths(run)
    assert torch.equal(new, private_a)
    ref = conv_ref(x, w, b, stride, x2, ups)
    if rowvec is not None:
        ref = ref + rowvec.float()[:, None, None, :]
    if res is not None:
        ref = ref + res.float()
    assert rel_err(new, ref) < 2e-3
    assert rel_err(same_order, ref) < 2e-3
    assert rel_err(new, same_order.float(), l2_tol=1e-4, row_tol=1e-3) < 2e-3      # two summation orders of the same products
    if nimg * Ho * Wo >= 8192:       # (smaller problems take split-K on the tile-kernel side: other summation order)
        assert torch.equal(old, same_order)
    else:
        assert rel_err(old, ref) < 2e-3


@pytest.mark.parametrize('nimg,Hs,Ws,C,Cout', [
    (32, 32, 32, 640, 640),        # Upsample3D 32 -> 64 at B = 2 (the 896-us launch of a forward)
    (16, 16, 16, 1280, 1280),      # 16 -> 32 at B = 1
    (16, 14, 24, 1280, 1280),      # the 448 x 768 clip's 14 x 24 -> 28 x 48 (W not a power of two)
    (32, 8, 8, 1280, 1280),        # 8 -> 16: too few tiles for the persistent kernel, stays on the nine-tap form
])
def test_subpixel_form_of_the_nearest_2x_convolution(nimg, Hs, Ws, C, Cout):
    """conv3x3(nearest_2x(x)) as four 2x2-tap convolutions of the source with the coinciding filter rows / columns added up
    (vsx_gemm_desc.upsample = 2, ABI 9; ops.subpixel_weights): 4/9 of the multiplications, one launch, the epilogue scatters
    every class to its pixels.  Against the fp32 reference, against the nine-tap kernel (the sums of fp16 filter taps are
    rounded to fp16 once: the two differ by that rounding), and the combined weights against their definition."""
    x = rnd(nimg, Hs, Ws, C, seed=160)
    w, b = rnd(Cout, 3, 3, C, seed=161, scale=(9 * C) ** -0.5), rnd(Cout, seed=162)
    o = ops()
    w4 = o.subpixel_weights(w).float()
    wf = w.float()
    assert torch.equal(w4[0, :, 0, 0], wf[:, 0, 0].half().float())                                   # class (0, 0): the corner alone
    assert torch.allclose(w4[0, :, 1, 1], wf[:, 1:, 1:].sum((1, 2)), atol=2e-3, rtol=2e-3)          # ... and the 2x2 block
    assert torch.allclose(w4[3, :, 1, 1], wf[:, :2, :2].sum((1, 2)), atol=2e-3, rtol=2e-3)          # class (1, 1)
    assert float(w4[0, :, 2].abs().max()) == 0 and float(w4[3, :, 0].abs().max()) == 0               # taps no class reads
    eligible = o._subpixel_eligible(nimg, Hs, Ws, C, Cout, 3, 1, None, None, None, None)
    assert eligible == (nimg * Hs * Ws >= 4096)
    sub = o.conv2d(x, w, b, upsample=True)
    o.CONV_SUBPIXEL = False
    try:
        nine = o.conv2d(x, w, b, upsample=True)
    finally:
        o.CONV_SUBPIXEL = True
    ref = conv_ref(x, w, b, 1, None, True)
    assert rel_err(sub, ref) < 2e-3
    if eligible:
        assert not torch.equal(sub, nine)
        assert rel_err(sub, nine.float(), l2_tol=6e-4, row_tol=3e-3) < 4e-3
    else:
        assert torch.equal(sub, nine)


@pytest.mark.parametrize('M,N,K,res', [(65536, 320, 320, True), (65536, 320, 320, False), (16384, 640, 640, True),
                                       (65500, 320, 328, True), (8192, 1280, 1280, True), (4096, 1280, 1280, True)])
def test_row_statistics_from_the_producing_gemm(M, N, K, res):
    """vsx_gemm_desc.rowstats (ABI 8): the persistent kernel's epilogue also writes (sum, sum of squares) of every output
    row over 6 column parts per 320 columns (the weight-stationary K = N = 320 kernel: 5 parts of 64 columns); vsx_row_stats_combine turns them into the (rstd, -rstd * mean) pairs the
    consumer of a folded LayerNorm reads.  Checked against the partial sums of the stored output, against vsx_row_stats
    on that output, and end to end: LayerNorm -> Linear from the producer's statistics equals the standalone-pass form.
    Launches the persistent kernel does not take (M = 4096: tile kernels) must simply not offer statistics."""
    import ctypes
    from videoswap_amd import _lib
    x, w, b = rnd(M, K, seed=150), rnd(N, K, seed=151, scale=K ** -0.5), rnd(N, seed=152)
    r = rnd(M, N, seed=153, scale=3.0) if res else None
    y = ops().linear(x, w, b, residual=r, row_stats=True)
    plain = ops().linear(x, w, b, residual=r)
    assert torch.equal(y, plain), 'emitting the statistics must not change the output'
    parts = getattr(y, '_vsx_rowparts', None)
    if M == 4096:
        assert parts is None
        return
    ws = N == 320 and K == 320 and res and M % 32 == 0 and M >= 65536      # the weight-stationary kernel (round 6): a part per wave = 32 columns
    assert parts is not None and parts.shape == (M, 10 if ws else (N // 320) * 6, 2)
    cols = []
    for h in range(N // 160):
        cols += [(h * 160, 64), (h * 160 + 64, 64), (h * 160 + 128, 32)]
    if ws:
        cols = [(32 * i, 32) for i in range(10)]
    yf = y.float()
    want = torch.stack([torch.stack([yf[:, c0:c0 + wd].sum(1), (yf[:, c0:c0 + wd] ** 2).sum(1)], -1) for c0, wd in cols], 1)
    assert (parts - want).abs().max() <= 1e-3 * (1 + want.abs().max())
    lib = _lib.load()
    st = torch.empty(M, 2, dtype=torch.float32, device=y.device)
    _lib.check(lib.vsx_row_stats_combine(ops()._p(parts), M, parts.shape[1], N, 1e-5, ops()._p(st), ops()._stream()), 'combine')
    ref = torch.empty_like(st)
    _lib.check(lib.vsx_row_stats(ops()._p(y), M, N, 1e-5, ops()._p(ref), ops()._stream()), 'row_stats')
    assert (st[:, 0] / ref[:, 0] - 1).abs().max() < 1e-4                         # rstd
    assert (st[:, 1] - ref[:, 1]).abs().max() < 1e-3 * (1 + ref[:, 1].abs().max())      # -rstd * mean
    # end to end through the fold: LayerNorm(y) @ w2^T with the statistics taken from the producer / from the pass over y
    from videoswap_amd.layers import LayerNorm
    ln = LayerNorm(N).to(y.device, torch.float16)
    with torch.no_grad():
        ln.weight.copy_(rnd(N, seed=154).abs() + 0.5)
        ln.bias.copy_(rnd(N, seed=155))
    w2 = rnd(320, N, seed=156, scale=N ** -0.5)
    a = ops().linear(ln(y, defer=True), w2)
    bare = y.clone()                                  # (a clone carries no statistics)
    b2 = ops().linear(ln(bare, defer=True), w2)
    ref2 = F.layer_norm(y.float(), (N,), ln.weight.float(), ln.bias.float(), 1e-5) @ w2.float().t()
    assert rel_err(a, ref2) < 2e-3 and rel_err(b2, ref2) < 2e-3
    assert rel_err(a, b2.float(), l2_tol=2e-4, row_tol=2e-3) < 2e-3


def test_persistent_kernel_is_deterministic_and_repeatable():
    """Back-to-back launches on one stream reuse the LDS ring and the barrier pattern: 20 launches, identical bits."""
    x, w, b = rnd(65536, 320, seed=103), rnd(960, 320, seed=104, scale=320 ** -0.5), rnd(960, seed=105)
    ops().set_option('gemm_pp', 2)
    try:
        first = ops().linear(x, w, b)
        for _ in range(20):
            assert torch.equal(ops().linear(x, w, b), first)
    finally:
        ops().set_option('gemm_pp', 1)


def test_conv2d_epilogue_rowvec_residual():
    nimg, H, W, C, Cout = 4, 8, 8, 64, 128
    x, w, b = rnd(nimg, H, W, C, seed=15), rnd(Cout, 3, 3, C, seed=16, scale=(9 * C) ** -0.5), rnd(Cout, seed=17)
    rowvec = rnd(2, Cout, seed=18)           # 2 "batches" of 2 frames
    res = rnd(nimg, H, W, Cout, seed=19)
    out = ops().conv2d(x, w, b, rowvec=rowvec, rows_per_vec=2 * H * W, residual=res)
    ref = conv_ref(x, w, b, 1) + rowvec.float().repeat_interleave(2, 0)[:, None, None, :] + res.float()
    assert rel_err(out, ref) < 2e-3


# --------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize('nimg,rows,C1,C2,groups,silu', [
    (2, 4 * 64, 64, 0, 32, True),       # 5-D statistics: B=2, F*HW rows
    (8, 64, 320, 0, 32, False),        # per-frame
    (2, 1000, 640, 320, 32, True),     # concat, cpg = 30 (vector straddles groups)
    (2, 700, 1280, 1280, 32, True),    # C = 2560
    (3, 50, 32, 0, 8, False),
])
def test_group_norm(nimg, rows, C1, C2, groups, silu):
    x = rnd(nimg, rows, C1, seed=20) + 0.5
    x2 = rnd(nimg, rows, C2, seed=21) if C2 else None
    C = C1 + C2
    gamma, beta = rnd(C, seed=22) + 1.0, rnd(C, seed=23)
    out = ops().group_norm(x, gamma, beta, groups, 1e-5, nimg, silu=silu, x2=x2)
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    ref = F.group_norm(xin.transpose(1, 2), groups, gamma.float(), beta.float(), 1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    assert rel_err(out, ref) < 2e-3
    # few chunks per image (the per-frame GroupNorms): the apply kernel can finalize the statistics itself (option gn_fuse = 1, round 6:
    # one launch fewer; measured 2 % slower and left off) — same bits as the stand-alone finalize kernel
    try:
        ops().set_option('gn_fuse', 1)
        fused = ops().group_norm(x, gamma, beta, groups, 1e-5, nimg, silu=silu, x2=x2)
    finally:
        ops().set_option('gn_fuse', 0)
    assert torch.equal(out, fused)


@pytest.mark.parametrize('nimg,rows,C', [(16, 4096, 320), (32, 4096, 320), (64, 1024, 640), (32, 256, 1280), (128, 64, 1280), (1, 65536, 320)])
def test_group_norm_fused_finalize_at_the_model_shapes(nimg, rows, C):
    """The GroupNorms in front of `proj_in` (attention.py:61,110; motion_module.py:112,149) at the UNet's real shapes — per frame, 13 - 49
    chunks per image: statistics optionally finalized inside the apply kernel (gn_fuse = 1) — and the 5-D GroupNorm of a resnet at B = 1
    (771 chunks: always the stand-alone finalize kernel): against PyTorch, and fused == unfused bit for bit."""
    x = rnd(nimg, rows, C, seed=320) * 1.5 + 0.3
    gamma, beta = rnd(C, seed=321) + 1.0, rnd(C, seed=322)
    out = ops().group_norm(x, gamma, beta, 32, 1e-6, nimg)
    ref = F.group_norm(x.float().transpose(1, 2), 32, gamma.float(), beta.float(), 1e-6).transpose(1, 2)
    assert rel_err(out, ref) < 2e-3
    try:
        ops().set_option('gn_fuse', 1)
        fused = ops().group_norm(x, gamma, beta, 32, 1e-6, nimg)
    finally:
        ops().set_option('gn_fuse', 0)
    assert torch.equal(out, fused)


@pytest.mark.parametrize('M,C', [(1000, 320), (64, 1280), (7, 64), (33, 640)])
def test_layer_norm(M, C):
    x, g, b = rnd(M, C, seed=24) * 2 + 0.3, rnd(C, seed=25) + 1, rnd(C, seed=26)
    out = ops().layer_norm(x, g, b, 1e-5)
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5)
    assert rel_err(out, ref) < 2e-3


def test_layer_norm_pe():
    B, Fr, HW, C = 2, 4, 16, 64
    x, g, b, pe = rnd(B * Fr * HW, C, seed=27), rnd(C, seed=28) + 1, rnd(C, seed=29), rnd(24, C, seed=30)
    out = ops().layer_norm(x, g, b, 1e-5, pe=pe, rows_per_frame=HW, frames=Fr, frame_offset=2)
    ref = F.layer_norm(x.float(), (C,), g.float(), b.float(), 1e-5).view(B, Fr, HW, C)
    ref = ref + pe.float()[2:2 + Fr][None, :, None, :]
    assert rel_err(out, ref.reshape(-1, C)) < 2e-3


# --------------------------------------------------------------------------------------------
# LayerNorm folded into the consuming GEMM (vsx_gemm_desc.rowscale / colvec, ops.DeferredLN): every epilogue that can
# receive it — persistent staged rows (plain, GEGLU), tile kernels (16-byte stores, GEGLU, generic), the transposed V^T
# store, the split-K combine — against LayerNorm -> Linear in fp32
# --------------------------------------------------------------------------------------------
def _ln_inputs(M, K, seed):
    # rows with very different means and scales: the identity subtracts rstd * mean * sum_k W'
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * (0.5 + 3 * torch.rand(M, 1, generator=g)) + 4 * torch.randn(M, 1, generator=g)
    return x.half().to(DEV), rnd(K, seed=seed + 1) * 0.3 + 1, rnd(K, seed=seed + 2) * 0.2


@pytest.mark.parametrize('M,K,N,geglu,res,pe', [
    (65536, 320, 960, False, False, True),     # persistent 256-row tiles, temporal qkv with the positional encoding
    (65536, 320, 320, False, True, False),     # persistent, residual
    (32768, 320, 1280, True, False, False),    # persistent GEGLU
    (8192, 1280, 1280, False, True, False),    # 128-row persistent tiles
    (4096, 640, 640, False, False, False),     # tile kernel, 16-byte stores
    (1000, 1280, 5120, True, False, False),    # tile kernel GEGLU, ragged M
    (1024, 1280, 3840, False, False, False),   # split-K combine
    (77, 768, 320, False, False, False),       # 64x64 tiles, ragged
])
def test_layer_norm_folded_into_linear(M, K, N, geglu, res, pe):
    o = ops()
    x, gamma, beta = _ln_inputs(M, K, seed=40)
    w, b = rnd((2 * N if geglu else N), K, seed=43, scale=K ** -0.5), rnd((2 * N if geglu else N), seed=44)
    r = rnd(M, N, seed=45) if res else None
    frames, hw = 4, M // 8 if pe else 0
    table = rnd(24, K, seed=46) if pe else None
    ln = o.DeferredLN(x, gamma, beta, 1e-5, table, hw, frames, 2 if pe else 0)
    out = o.linear(ln, w, b, residual=r, geglu=geglu)
    y = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
    if pe:
        y = (y.view(-1, frames, hw, K) + table.float()[2:2 + frames][None, :, None, :]).view(M, K)
    ref = y @ w.float().t() + b.float()
    if geglu:
        ref = ref[:, :N] * F.gelu(ref[:, N:])
    if res:
        ref = ref + r.float()
    assert rel_err(out, ref) < 2e-3
    # the same call through the materialised LayerNorm (what VSX_LN_FUSE=0 runs) agrees to fp16 noise
    plain = o.linear(ln.materialize(), w, b, residual=r, geglu=geglu)
    assert rel_err(out, plain.float(), l2_tol=2e-3, row_tol=1e-2) < 4e-3


@pytest.mark.parametrize('nimg,rows,K,N', [(16, 4096, 320, 320), (32, 256, 1280, 1280), (3, 77, 768, 640)])
def test_layer_norm_folded_into_the_transposed_v_projection(nimg, rows, K, N):
    o = ops()
    x, gamma, beta = _ln_inputs(nimg * rows, K, seed=50)
    w, b = rnd(N, K, seed=53, scale=K ** -0.5), rnd(N, seed=54)
    ln = o.DeferredLN(x, gamma, beta, 1e-5)
    vt = o.linear_vt(ln, w, b, rows)
    y = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5) @ w.float().t() + b.float()
    ref = y.view(nimg, rows, N).transpose(1, 2)
    assert rel_err(vt[:, :, :rows], ref) < 2e-3


def test_row_stats_are_computed_once_per_deferred_layer_norm():
    o = ops()
    x, gamma, beta = _ln_inputs(2048, 320, seed=60)
    ln = o.DeferredLN(x.view(2, 1024, 320), gamma, beta, 1e-5)
    st = ln.stats()
    assert ln.reshape(2048, 320).stats() is st and ln.view(2048, 320).stats() is st
    mean, var = x.float().mean(1), x.float().var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0], rstd, rtol=1e-4) and torch.allclose(st[:, 1], -rstd * mean, rtol=1e-4, atol=1e-5)


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def attn_ref(q, k, v, heads, scale, kv_div=1):
    nb, nq, C = q.shape
    d = C // heads
    qh = q.float().view(nb, nq, heads, d).transpose(1, 2)
    kh = k.float().view(k.shape[0], -1, heads, d).transpose(1, 2).repeat_interleave(kv_div, 0)
    vh = v.float().view(v.shape[0], -1, heads, d).transpose(1, 2).repeat_interleave(kv_div, 0)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(nb, nq, C), p


def make_vt(v):
    """[nkvb, nk, C] -> V^T [nkvb, C, round_up(nk, 8)] (zero padded)"""
    nkvb, nk, C = v.shape
    ld = (nk + 7) // 8 * 8
    vt = torch.zeros(nkvb, C, ld, dtype=v.dtype, device=v.device)
    vt[:, :, :nk] = v.transpose(1, 2)
    return vt


@pytest.mark.parametrize('nb,heads,nq,nk,d', [
    (2, 8, 256, 256, 40), (2, 8, 128, 128, 80), (2, 4, 64, 64, 160), (1, 2, 1024, 1024, 40),
    (2, 8, 84, 84, 40), (1, 8, 336, 336, 80), (2, 2, 200, 200, 8), (2, 2, 64, 64, 16), (2, 2, 100, 130, 32),
    (1, 2, 128, 192, 64), (1, 1, 64, 64, 128),
])
def test_attention_self(nb, heads, nq, nk, d):
    C = heads * d
    q, k, v = rnd(nb, nq, C, seed=31), rnd(nb, nk, C, seed=32), rnd(nb, nk, C, seed=33)
    scale = d ** -0.5
    out = ops().attention(q, k, make_vt(v), heads, scale)
    ref, _ = attn_ref(q, k, v, heads, scale)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3


@pytest.mark.parametrize('qb', [1, 2])
@pytest.mark.parametrize('nb,heads,nq,nk,d,kv_div', [(2, 8, 300, 300, 40, 1), (2, 8, 1024, 1001, 40, 1), (4, 8, 260, 77, 80, 2),
                                                     (1, 8, 336, 336, 80, 1)])
def test_attention_query_blocks_per_wave(qb, nb, heads, nq, nk, d, kv_div):
    """32 / 64 queries per wave (option attn_qb; the launch rule takes 64 only for the big self-attention launches, so both
    are forced here): ragged query tiles (300 = 256 + 44), a ragged key count (1001: the peeled partial tile), shared text
    K / V.  The two forms must also agree with each other far inside the tolerance (same arithmetic per query)."""
    C = heads * d
    q = rnd(nb, nq, C, seed=131)
    k, v = rnd(nb // kv_div, nk, C, seed=132), rnd(nb // kv_div, nk, C, seed=133)
    scale = d ** -0.5
    ref, _ = attn_ref(q, k, v, heads, scale, kv_div=kv_div)
    try:
        ops().set_option('attn_o16', 0)         # both forms on 32-row O^T tiles (the default; the optional 16-row tiles sum the keys of
        ops().set_option('attn_qb', qb)         # a block in another order: test_attention_o_tiles_of_16_rows)
        out = ops().attention(q, k, make_vt(v), heads, scale, kv_div=kv_div)
        ops().set_option('attn_qb', 3 - qb)
        other = ops().attention(q, k, make_vt(v), heads, scale, kv_div=kv_div)
    finally:
        ops().set_option('attn_qb', 0)
        ops().set_option('attn_o16', 0)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3
    assert torch.equal(out, other), 'a query\'s result must not depend on how many query blocks its wave owns'


@pytest.mark.parametrize('nb,heads,nq,nk,kv_div', [(2, 8, 300, 300, 1), (2, 8, 1024, 1001, 1), (4, 8, 1024, 77, 2), (1, 8, 4096, 4096, 1),
                                                   (2, 5, 96, 333, 1)])
def test_attention_o_tiles_of_16_rows(nb, heads, nq, nk, kv_div):
    """d = 40 (round 6; option attn_o16 = 1, measured and not made the default): O^T on 16 x 16 x 32 MFMA tiles — 40 channel rows padded to 48 instead of 64, P^T brought into the B-operand
    layout by v_permlane16_swap, the rescale factors redistributed the same way, the softmax denominator from the ones row 47
    (csrc/attention.hip, O16; tools/ubench/mfma16_probe.hip probed both instructions).  Against the fp32 reference under the same
    tolerance as every attention test, and against the 32-row form (option attn_o16 = 0): the two sum a key block's products in
    another order, so they agree to fp32 rounding of the accumulation, not bit for bit.  Shapes: ragged query tiles, the peeled
    partial key tile (1001, 333), text keys shared by two batches, the 64 x 64 level's 4096 x 4096, five heads."""
    d = 40
    C = heads * d
    q = rnd(nb, nq, C, seed=231)
    k, v = rnd(nb // kv_div, nk, C, seed=232), rnd(nb // kv_div, nk, C, seed=233)
    if nk > 200:
        k[:, 150] = q[: nb // kv_div, 7] * 3.0          # a late dominant key: the rescale path (alpha through permlane16_swap)
    scale = d ** -0.5
    ref, _ = attn_ref(q, k, v, heads, scale, kv_div=kv_div)
    try:
        ops().set_option('attn_o16', 1)
        out = ops().attention(q, k, make_vt(v), heads, scale, kv_div=kv_div)
        again = ops().attention(q, k, make_vt(v), heads, scale, kv_div=kv_div)
        ops().set_option('attn_o16', 0)
        old = ops().attention(q, k, make_vt(v), heads, scale, kv_div=kv_div)
    finally:
        ops().set_option('attn_o16', 0)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < (6e-3 if nq >= 4096 else 4e-3)
    assert torch.equal(out, again)
    e_new, e_old = rel_err(out, ref, l2_tol=1.0, row_tol=1.0), rel_err(old, ref, l2_tol=1.0, row_tol=1.0)
    assert e_new <= 1.1 * e_old + 1e-5, (e_new, e_old)
    assert float((out.float() - old.float()).norm() / old.float().norm()) < 5e-4


@pytest.mark.parametrize('nb,heads,nq,nk,d,kv_div', [(2, 8, 1024, 1024, 40, 1), (2, 8, 1100, 1100, 40, 1), (4, 8, 1024, 77, 40, 2),
                                                     (2, 8, 400, 400, 80, 1), (1, 2, 256, 256, 64, 1)])
def test_attention_backward_flash(nb, heads, nq, nk, d, kv_div):
    """vsx_attention_lse_f16 + vsx_attention_bwd_f16 (csrc/attention_bwd.hip) against PyTorch autograd of the fp32
    softmax(scale Q K^T) V on the same fp16 inputs: output, log-sum-exp, dQ, dK, dV (shared text K / V: dQ only); ragged
    query / key tiles (1100 = 8 x 128 + 76)."""
    C = heads * d
    q, k, v = rnd(nb, nq, C, seed=141), rnd(nb // kv_div, nk, C, seed=142), rnd(nb // kv_div, nk, C, seed=143)
    g = rnd(nb, nq, C, seed=144)
    scale = d ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref, _ = attn_ref(qr, kr, vr, heads, scale, kv_div=kv_div)
    (ref * g.float()).sum().backward()
    out, lse = ops().attention_lse(q, k, make_vt(v), heads, scale, kv_div=kv_div)
    assert rel_err(out, ref.detach(), l2_tol=3e-3, row_tol=1.2e-2) < 4e-3
    qh = q.float().view(nb, nq, heads, d).transpose(1, 2)
    kh = k.float().view(nb // kv_div, nk, heads, d).transpose(1, 2).repeat_interleave(kv_div, 0)
    want_lse = torch.logsumexp(qh @ kh.transpose(-1, -2) * scale, -1) * 1.4426950408889634
    assert (lse[..., :nq] - want_lse).abs().max() < 2e-2 and float(lse[..., nq:].abs().sum()) == 0.0
    self_attn = kv_div == 1
    dq, dk, dv = ops().attention_bwd(q, k, v, out, g, lse, heads, scale, kv_div=kv_div, need_kv=self_attn)
    assert rel_err(dq, qr.grad, l2_tol=4e-3, row_tol=2e-2) < 8e-3
    if self_attn:
        assert rel_err(dk, kr.grad, l2_tol=4e-3, row_tol=2e-2) < 8e-3
        assert rel_err(dv, vr.grad, l2_tol=4e-3, row_tol=2e-2) < 8e-3
    else:
        assert dk is None and dv is None


def test_attention_self_benchmark_shape():
    """N = 4096, d = 40, 8 heads: the 64x64-level self-attention of the benchmarked model (32 query tiles per head,
    XCD-ordered workgroups); nb = 4 keeps the fp32 reference's score tensor at 2 GiB."""
    nb, heads, nq, d = 4, 8, 4096, 40
    C = heads * d
    q, k, v = rnd(nb, nq, C, seed=83), rnd(nb, nq, C, seed=84), rnd(nb, nq, C, seed=85)
    out = ops().attention(q, k, make_vt(v), heads, d ** -0.5)
    ref, _ = attn_ref(q, k, v, heads, d ** -0.5)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 6e-3


def test_attention_peaked_softmax():
    # forces the online-softmax rescale: one key dominates late in the key sequence
    nb, heads, nq, nk, d = 1, 2, 128, 256, 40
    C = heads * d
    q, k, v = rnd(nb, nq, C, seed=34), rnd(nb, nk, C, seed=35), rnd(nb, nk, C, seed=36)
    k[:, 200] = q[:, 5] * 4.0
    k[:, 70] = q[:, 9] * 3.0
    scale = d ** -0.5
    out = ops().attention(q, k, make_vt(v), heads, scale)
    ref, _ = attn_ref(q, k, v, heads, scale)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3


@pytest.mark.parametrize('frames,nq,d', [(4, 256, 40), (2, 64, 160), (3, 100, 80)])
def test_attention_cross_text(frames, nq, d):
    B, heads, nk = 2, 8, 77
    C = heads * d
    q = rnd(B * frames, nq, C, seed=37)
    k, v = rnd(B, nk, C, seed=38), rnd(B, nk, C, seed=39)
    scale = d ** -0.5
    out = ops().attention(q, k, make_vt(v), heads, scale, kv_div=frames)
    ref, _ = attn_ref(q, k, v, heads, scale, kv_div=frames)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3


@pytest.mark.parametrize('nb,heads,nq,nk,d,kv_div', [(4, 8, 64, 64, 40, 1), (4, 8, 256, 77, 40, 2), (2, 2, 100, 100, 16, 1)])
def test_attention_scores_and_pv(nb, heads, nq, nk, d, kv_div):
    C = heads * d
    q = rnd(nb, nq, C, seed=40)
    k, v = rnd(nb // kv_div, nk, C, seed=41), rnd(nb // kv_div, nk, C, seed=42)
    scale = d ** -0.5
    probs = ops().attention_scores(q, k, heads, scale, kv_div=kv_div)
    ref_out, ref_p = attn_ref(q, k, v, heads, scale, kv_div=kv_div)
    assert probs.shape == ref_p.shape
    assert rel_err(probs, ref_p, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3
    out = ops().attention_pv(probs, make_vt(v), kv_div=kv_div)
    assert rel_err(out, ref_out, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3
    # an edited (non-view) probs tensor must work too
    out2 = ops().attention_pv((probs * 1.0).contiguous(), make_vt(v), kv_div=kv_div)
    assert rel_err(out2, ref_out, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3


@pytest.mark.parametrize('B,fq,fk,hw,heads,d', [(2, 16, 16, 64, 8, 40), (1, 4, 4, 256, 8, 80), (2, 16, 16, 16, 8, 160),
                                               (2, 16, 16, 4096, 8, 40), (1, 8, 8, 1024, 8, 80),
                                               # long-clip mode: local query frames x gathered key frames (MFMA kernel)
                                               (1, 16, 64, 256, 8, 40), (2, 16, 64, 64, 8, 80), (1, 16, 64, 16, 8, 160),
                                               (1, 8, 40, 50, 8, 40), (1, 24, 96, 10, 8, 80), (1, 16, 128, 8, 6, 40),
                                               (1, 4, 16, 30, 2, 8), (1, 24, 24, 10, 4, 16)])
def test_temporal_attention(B, fq, fk, hw, heads, d):
    C = heads * d
    q, k, v = rnd(B * fq * hw, C, seed=43), rnd(B * fk * hw, C, seed=44), rnd(B * fk * hw, C, seed=45)
    scale = d ** -0.5
    out = ops().temporal_attention(q, k, v, B, fq, fk, hw, heads, scale)

    def sites(t, f):  # (b f s) c -> (b s) f c
        return t.float().view(B, f, hw, C).permute(0, 2, 1, 3).reshape(B * hw, f, C)
    ref, _ = attn_ref(sites(q, fq), sites(k, fk), sites(v, fk), heads, scale)
    ref = ref.view(B, hw, fq, C).permute(0, 2, 1, 3).reshape(B * fq * hw, C)
    assert rel_err(out, ref, l2_tol=3e-3, row_tol=1.2e-2) < 3e-3


# --------------------------------------------------------------------------------------------
# element-wise glue
# --------------------------------------------------------------------------------------------
def test_silu_axpy():
    x, y = rnd(1000, 33, seed=46) * 3, rnd(1000, 33, seed=47)
    assert rel_err(ops().silu(x), F.silu(x.float())) < 1e-3
    assert rel_err(ops().axpy(x, y, 0.5), x.float() + 0.5 * y.float()) < 1e-3


def test_pack_unpack_latents():
    x = rnd(2, 4, 3, 6, 10, seed=48)
    p = ops().pack_latents(x, 8)
    ref = torch.zeros(6, 6, 10, 8, device=DEV)
    ref[..., :4] = x.float().permute(0, 2, 3, 4, 1).reshape(6, 6, 10, 4)
    assert torch.equal(p.float(), ref)
    y = rnd(6, 6, 10, 8, seed=49)
    u = ops().unpack_latents(y, 2, 4)
    assert torch.equal(u, y[..., :4].view(2, 3, 6, 10, 4).permute(0, 4, 1, 2, 3))


def test_cfg_ddim_step():
    x, eu, ec = rnd(1, 4, 4, 16, 16, seed=50), rnd(1, 4, 4, 16, 16, seed=51), rnd(1, 4, 4, 16, 16, seed=52)
    a_t, a_n, g = 0.37, 0.52, 7.5
    out = ops().cfg_ddim_step(x, eu, ec, g, a_t, a_n)
    e = eu.float() + g * (ec.float() - eu.float())
    x0 = (x.float() - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    ref = math.sqrt(a_n) * x0 + math.sqrt(1 - a_n) * e
    assert rel_err(out, ref) < 1e-3
    out1 = ops().cfg_ddim_step(x, eu, None, 1.0, a_t, a_n)
    x0 = (x.float() - math.sqrt(1 - a_t) * eu.float()) / math.sqrt(a_t)
    assert rel_err(out1, math.sqrt(a_n) * x0 + math.sqrt(1 - a_n) * eu.float()) < 1e-3


def test_masked_blend():
    x, s = rnd(4, 3, 8, 8, seed=53), rnd(4, 3, 8, 8, seed=54)
    m = (torch.rand(3, 8, 8, device=DEV) > 0.5).half()
    out = ops().masked_blend(x, s, m)
    assert rel_err(out, s.float() + m.float() * (x.float() - s.float())) < 1e-3


def test_adapter_scatter():
    Fr, P, C, h, w, rate = 3, 5, 64, 8, 12, 8.0
    g = torch.Generator().manual_seed(55)
    tracks = torch.rand(Fr, P, 2, generator=g) * torch.tensor([w * rate, h * rate])
    tracks[0, 1] = -1.0                      # invisible
    tracks[1, 2] = torch.tensor([w * rate - 0.5, h * rate - 0.5])  # clamps to the edge: corners coincide
    tracks[2, 3] = torch.tensor([16.0, 24.0])  # exactly on the grid
    tracks = tracks.half().float()
    sel = torch.tensor([1, 1, 1, 1, 0], dtype=torch.int32)
    feat = rnd(P, C, seed=56)
    out = ops().adapter_scatter(tracks.to(DEV), sel.to(DEV), feat, h, w, rate)
    ref = torch.zeros(Fr, h, w, C)
    fc = feat.float().cpu()
    for p in range(P):
        if not sel[p]:
            continue
        for f in range(Fr):
            px, py = tracks[f, p].tolist()
            if px < 0 or py < 0:
                continue
            x, y = px / rate, py / rate
            x1, y1 = int(x), int(y)
            x2, y2 = x1 + 1, y1 + 1
            xf, yf = x - x1, y - y1
            x1, x2 = max(min(x1, w - 1), 0), max(min(x2, w - 1), 0)
            y1, y2 = max(min(y1, h - 1), 0), max(min(y2, h - 1), 0)
            ref[f, y1, x1] += fc[p] * (1 - xf) * (1 - yf)
            ref[f, y1, x2] += fc[p] * xf * (1 - yf)
            ref[f, y2, x1] += fc[p] * (1 - xf) * yf
            ref[f, y2, x2] += fc[p] * xf * yf
    assert rel_err(out.cpu(), ref, l2_tol=3e-3, row_tol=1.2e-2) < 4e-3
    # on-grid point: a single pixel carries exactly the feature vector
    assert torch.equal(out[2, 3, 2].cpu(), feat[3].cpu()) or rel_err(out[2, 3, 2].cpu(), fc[3]) < 2e-3


def test_errors_are_reported():
    from videoswap_amd._lib import VsxError
    x = rnd(16, 12, seed=57)  # K = 12 is not a multiple of 8
    w = rnd(8, 12, seed=58)
    with pytest.raises(VsxError):
        ops().linear(x, w)
    with pytest.raises(VsxError):
        ops().linear(x.cpu(), w.cpu())


@pytest.mark.parametrize('waves', [10, 5])
@pytest.mark.parametrize('M', [131072, 32 * 1027, 96])
@pytest.mark.parametrize('res,stats', [(False, False), (True, False), (False, True), (True, True)])
def test_weight_stationary_320_matches_the_persistent_kernel(M, res, stats, waves):
    """gemm_ws320_kernel (csrc/gemm_pp.hip; option gemm_ws): the K = N = 320 projections of the 64 x 64 level with the weights in
    registers and the activation streamed in 32-row blocks.  Same instruction, operand roles and k order as the other back ends and
    the same staged row passes, so the output must agree BIT FOR BIT with the product's dispatch (and with fp32 to fp16 rounding);
    its row statistics come in one part per wave (option ws_waves: 10 waves of 32 columns, or 5 of 64) instead of 6 and must add up to the
    same sums.  M = 32 * 1027: workgroups with 5 and 4 blocks (the out-of-range tail pieces); M = 96: fewer blocks than CUs."""
    from videoswap_amd import ops
    ops.set_option('ws_waves', waves)
    g = torch.Generator(device=DEV).manual_seed(M % 1000 + 2 * res + stats)
    x = torch.randn(M, 320, device=DEV, generator=g).half()
    w = (torch.randn(320, 320, device=DEV, generator=g) * 320 ** -0.5).half()
    b = torch.randn(320, device=DEV, generator=g).half()
    r = torch.randn(M, 320, device=DEV, generator=g).half() if res else None
    outs, parts = [], []
    try:
        for v in (0, 2):
            ops.set_option('gemm_ws', v)
            y = ops.linear(x, w, b, residual=r, row_stats=stats)
            outs.append(y.clone())
            parts.append(getattr(y, '_vsx_rowparts', None))
    finally:
        ops.set_option('gemm_ws', 1)
        ops.set_option('ws_waves', 10)
    assert torch.equal(outs[0], outs[1])
    ref = x.float() @ w.float().t() + b.float() + (r.float() if res else 0.0)
    assert float((outs[1].float() - ref).norm() / ref.norm()) < 6e-4
    if stats:
        assert parts[1] is not None and parts[1].shape == (M, waves, 2)
        y = outs[1].float()
        want = torch.stack([y.sum(1), (y * y).sum(1)], 1)
        got = parts[1].sum(1)
        assert float((got - want).abs().max() / want.abs().max()) < 1e-5
        if parts[0] is not None:        # the persistent kernel's 6 parts of the same rounded outputs
            assert float((parts[0].sum(1) - got).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize('waves', [10, 5])
@pytest.mark.parametrize('M,N,pe', [(65536, 640, False), (65536, 960, True), (32 * 515, 960, True), (4096, 320, False)])
def test_weight_stationary_column_slices_with_the_folded_layernorm(M, N, pe, waves):
    """The same kernel over column slices of 320 (N = 640 / 960, K = 320) with the LayerNorm folded into the GEMM and the temporal positional
    row vector (gathered by LDS-DMA from the one or two vectors a 32-row block meets): bit for bit like the product's dispatch, and right
    against the explicit LayerNorm -> Linear in fp32.  M = 32 * 515: chains of unequal length; rows_per_frame = 515 * 2: blocks that
    straddle two vectors."""
    from videoswap_amd import ops
    g = torch.Generator(device=DEV).manual_seed(N + M % 97)
    x = (torch.randn(M, 320, device=DEV, generator=g) * 1.5 + 0.3).half()
    gam, bet = torch.randn(320, device=DEV, generator=g).half(), torch.randn(320, device=DEV, generator=g).half()
    w = (torch.randn(N, 320, device=DEV, generator=g) * 320 ** -0.5).half()
    b = torch.randn(N, device=DEV, generator=g).half()
    frames, rpf = 16, M // 16
    pet = torch.randn(24, 320, device=DEV, generator=g).half() if pe else None
    kw = dict(pe=pet, rows_per_frame=rpf, frames=frames) if pe else {}
    outs = []
    ops.set_option('ws_waves', waves)
    try:
        for v in (0, 2):
            ops.set_option('gemm_ws', v)
            outs.append(ops.linear(ops.DeferredLN(x, gam, bet, 1e-5, **kw), w, b).clone())
    finally:
        ops.set_option('gemm_ws', 1)
        ops.set_option('ws_waves', 10)
    assert torch.equal(outs[0], outs[1])
    xn = F.layer_norm(x.float(), (320,), gam.float(), bet.float(), 1e-5)
    if pe:
        xn = xn + pet[:frames].float().repeat_interleave(rpf, 0)
    ref = xn @ w.float().t() + b.float()
    assert float((outs[1].float() - ref).norm() / ref.norm()) < 2e-3
