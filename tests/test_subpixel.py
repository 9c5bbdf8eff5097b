"""Host side of the sub-pixel form of the nearest-2x convolution (include/vsx.h: vsx_gemm_desc.upsample = 2; reference:
Upsample3D, /root/reference/videoswap/models/animatediff_models/resnet.py:54,66 = F.interpolate(scale_factor=2, mode="nearest")
followed by a 3x3 convolution).  No GPU: the combined weights against their definition, the cache, the eligibility rule."""
import torch
import torch.nn.functional as F

from videoswap_amd import ops


def _scatter_conv(x, w4):
    """what the kernel computes: class (ph, pw) = a pad-1 3x3 window on the SOURCE with the class's weight matrix (five of its
    nine taps are zero), written to the output pixels (2 i + ph, 2 j + pw).  x [N, H, W, C], w4 [4, O, 3, 3, C]"""
    n, hs, ws, _ = x.shape
    out = torch.zeros(n, 2 * hs, 2 * ws, w4.shape[1], dtype=torch.float64)
    xin = x.double().permute(0, 3, 1, 2)
    for ph in (0, 1):
        for pw in (0, 1):
            y = F.conv2d(xin, w4[2 * ph + pw].double().permute(0, 3, 1, 2), padding=1)
            out[:, ph::2, pw::2] = y.permute(0, 2, 3, 1)
    return out


def test_subpixel_weights_reproduce_the_convolution_of_the_upsampled_image():
    torch.manual_seed(0)
    x = torch.randn(2, 5, 6, 8).half()                  # NHWC source, odd / even sizes
    w = (torch.randn(7, 3, 3, 8) * 0.2).half()          # OHWI
    w4 = ops.subpixel_weights(w)
    assert w4.shape == (4, 7, 3, 3, 8) and w4.dtype == torch.float16
    ref = F.conv2d(F.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2.0, mode='nearest'),
                   w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    got = _scatter_conv(x, w4)
    # the sums of two / four fp16 taps are rounded to fp16 once: that rounding is the whole difference
    assert float((got - ref).norm() / ref.norm()) < 5e-4
    # the taps a class never reads are zero; the corner taps are the filter's own corners
    for cls, (ph, pw) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        for kh in range(3):
            for kw in range(3):
                live = kh in (ph, ph + 1) and kw in (pw, pw + 1)
                assert live or float(w4[cls, :, kh, kw].abs().max()) == 0.0
    assert torch.equal(w4[0, :, 0, 0], w[:, 0, 0]) and torch.equal(w4[3, :, 2, 2], w[:, 2, 2])


def test_subpixel_weights_are_cached_per_weight_version():
    w = torch.randn(4, 3, 3, 8).half()
    a = ops.subpixel_weights(w)
    assert ops.subpixel_weights(w) is a                 # same tensor object, same version: the cached copy
    w.mul_(2)                                           # a LoRA merge / load_state_dict bumps the version
    b = ops.subpixel_weights(w)
    assert b is not a and torch.allclose(b.float(), 2 * a.float(), atol=2e-3, rtol=2e-3)
    key = id(w)
    del w, a, b
    import gc
    gc.collect()
    assert key not in ops._subpixel_cache               # the entry lives exactly as long as its weight


def test_subpixel_eligibility_mirrors_the_library_rule(monkeypatch):
    ok = lambda nimg, hs, ws, c, o, **kw: ops._subpixel_eligible(nimg, hs, ws, c, o, kw.get('ks', 3), kw.get('stride', 1),  # noqa: E731
                                                                 kw.get('x2'), kw.get('rowvec'), kw.get('residual'), kw.get('padding'))
    assert ok(32, 32, 32, 640, 640) and ok(16, 16, 16, 1280, 1280)           # the UNet's 32 -> 64 and 16 -> 32 at B = 2 / 1
    assert ok(16, 14, 24, 1280, 1280)                                         # the 448 x 768 clip
    assert not ok(32, 8, 8, 1280, 1280)                                       # too few tiles for the persistent kernel
    assert not ok(32, 32, 32, 640, 512)                                       # VAE-like width: no 320-column tiles
    assert not ok(32, 32, 32, 72, 640)                                        # a K slab would straddle taps
    assert not ok(3, 30, 30, 640, 640)                                        # rows per class not a multiple of the tile
    assert not ok(32, 32, 32, 640, 640, stride=2) and not ok(32, 32, 32, 640, 640, ks=1)
    assert not ok(32, 32, 32, 640, 640, residual=object()) and not ok(32, 32, 32, 640, 640, padding=(0, 1))
    monkeypatch.setitem(ops._options, 'gemm_pp', 0)                           # tests / A-B runs switch the persistent kernel off
    assert not ok(32, 32, 32, 640, 640)
    monkeypatch.setitem(ops._options, 'gemm_pp', 1)
    monkeypatch.setitem(ops._options, 'tile_tune', 2)                         # ... or force a tile
    assert not ok(32, 32, 32, 640, 640)
    monkeypatch.setitem(ops._options, 'tile_tune', 0)
    monkeypatch.setattr(ops, 'CONV_SUBPIXEL', False)
    assert not ok(32, 32, 32, 640, 640)
