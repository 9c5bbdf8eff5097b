"""Host side of the LayerNorm fold (videoswap_amd/ops.py: DeferredLN, _ln_folded, _ln_pe_rows): the operands handed to
vsx_gemm_f16 (W o gamma, c1, c2, the pe W^T row vectors) reproduce LayerNorm -> (+ positional encoding) -> Linear
(attention.py:182-206, motion_module.py:213-219,253-255) through the epilogue identity of include/vsx.h.  No GPU: the GEMM and
the row statistics are restated in float64 here; the kernels' side is tests/test_kernels_gpu.py::test_layer_norm_folded_*."""
import torch

from videoswap_amd import ops


def _case(pe):
    g = torch.Generator().manual_seed(5)
    frames, rpf, C, N = 4, 8, 64, 48
    M = 3 * frames * rpf
    x = (torch.randn(M, C, generator=g) * 2.0 + 0.7).half()
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
    W, b = (torch.randn(N, C, generator=g) / 8).half(), (0.1 * torch.randn(N, generator=g)).half()
    pos = (0.5 * torch.randn(frames + 2, C, generator=g)).half() if pe else None
    ln = ops.DeferredLN(x, gamma, beta, 1e-5, pe=pos, rows_per_frame=rpf if pe else 0, frames=frames if pe else 0,
                        frame_offset=1 if pe else 0)
    return x, gamma, beta, W, b, pos, ln, M, frames, rpf


def _check(pe):
    x, gamma, beta, W, b, pos, ln, M, frames, rpf = _case(pe)
    hit = ops._ln_folded(W, b, ln)
    wf, c1, c2 = hit[2], hit[3], hit[4]
    assert wf.dtype == torch.float16 and c1.dtype == torch.float32 and c2.dtype == torch.float16
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rs = (var + 1e-5).rsqrt()
    rt = -rs * mean                                                  # what vsx_row_stats writes: (rstd, -rstd * mean)
    out = rs * (xd @ wf.double().t()) + rt * c1.double()[None, :] + c2.double()[None, :]
    y = (xd - mean) * rs * gamma.double() + beta.double()
    if pe:
        rv, rpv = ops._ln_pe_rows(hit, W, ln, M)
        assert rpv == rpf and rv.shape == (M // rpf, W.shape[0])
        rows = torch.arange(M) // rpv
        out = out + rv.double()[rows]
        y = y + pos.double()[1 + rows % frames]                      # frame_offset = 1: row block i takes frame i % frames
    ref = y @ W.double().t() + b.double()
    err = (out - ref).norm() / ref.norm()
    assert err < 1e-3, err                                           # fp16 rounding of W o gamma, c2 and the pe rows
    return hit


def test_fold_reproduces_layernorm_linear():
    _check(pe=False)


def test_fold_with_temporal_positional_encoding():
    _check(pe=True)


def test_fold_cache_follows_parameter_updates():
    x, gamma, beta, W, b, pos, ln, *_ = _case(False)
    a = ops._ln_folded(W, b, ln)
    assert ops._ln_folded(W, b, ln) is a                             # same parameters: cached operands
    W.mul_(2)                                                        # LoRA merge / load_state_dict: version bump
    c = ops._ln_folded(W, b, ln)
    assert c is not a and torch.allclose(c[3], 2 * a[3], rtol=2e-3)
