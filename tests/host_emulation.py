"""TEST INFRASTRUCTURE — a CPU stand-in for `videoswap_amd.ops`, so that the HOST side of the product (module wiring,
layouts, caches, the frame-sharded exchanges, pipeline bookkeeping) can be exercised in the `-m "not gpu"` suite.

It is NOT a fallback: nothing under videoswap_amd/ imports it, the product's ops raise on CPU tensors, and it lives
under tests/.  Each function below restates the CONTRACT of the op of the same name in videoswap_amd/ops.py (argument
layouts as documented there) in plain PyTorch: fp16 tensors in and out, fp32 arithmetic inside — the rounding points
of the kernels (one fp16 rounding per op output) but none of their code.  The kernels themselves are checked against
PyTorch references on the GPU (tests/test_kernels_gpu.py); what this file makes testable on CPU is everything AROUND
them.  Use `with host_emulation.installed():`; tests marked `device` get it from tests/conftest.py when there is no GPU."""
import contextlib
import functools
import math

import torch
import torch.nn.functional as F

H = torch.float16


def _f(t):
    return None if t is None else t.float()


def linear(x, weight, bias=None, residual=None, geglu=False, out=None, row_stats=False):      # (row_stats: a hint the stand-in ignores)
    w = weight.reshape(weight.shape[0], -1).float()
    y = x.float() @ w.t()
    if bias is not None:
        y = y + bias.float()
    if geglu:
        h, g = y.chunk(2, dim=-1)
        y = h * F.gelu(g)
    if residual is not None:
        y = y + residual.float().reshape(y.shape)
    y = y.to(H)
    if out is not None:
        out.copy_(y.view(out.shape))
        return out
    return y


def linear_vt(x, weight, bias, rows_per_img, ldvt=None):
    K = x.shape[-1]
    y = linear(x.reshape(-1, K), weight, bias)                       # [nimg*rows, N]
    N = y.shape[-1]
    nimg = y.shape[0] // rows_per_img
    ldvt = ldvt or (rows_per_img + 7) // 8 * 8
    vt = torch.zeros(nimg, N, ldvt, dtype=H)
    vt[:, :, :rows_per_img] = y.view(nimg, rows_per_img, N).transpose(1, 2)
    return vt


def conv2d(x, weight, bias=None, *, x2=None, stride=1, upsample=False, rowvec=None, rows_per_vec=0, residual=None,
           padding=None):
    xin = x if x2 is None else torch.cat([x, x2], dim=-1)
    xin = xin.float().permute(0, 3, 1, 2)                            # NCHW
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ks = weight.shape[1]
    lo, hi = (ks // 2, ks // 2) if padding is None else padding
    xin = F.pad(xin, (lo, hi, lo, hi))
    w = weight.float().permute(0, 3, 1, 2)                           # [O, kh, kw, I] -> OIHW
    y = F.conv2d(xin, w, None if bias is None else bias.float(), stride=stride).permute(0, 2, 3, 1)
    if rowvec is not None:
        n, ho, wo, co = y.shape
        rows = y.reshape(-1, co)
        idx = torch.arange(rows.shape[0]) // rows_per_vec
        y = (rows + rowvec.float()[idx]).view(n, ho, wo, co)
    if residual is not None:
        y = y + residual.float().reshape(y.shape)
    return y.to(H).contiguous()


def _heads(t, heads):
    b, n, c = t.shape
    return t.float().view(b, n, heads, c // heads).permute(0, 2, 1, 3)        # [b, h, n, d]


def attention_scores(q, k, heads, scale, kv_div=1, causal=False, softmax=True):
    qh = _heads(q, heads)
    kh = _heads(k, heads).repeat_interleave(kv_div, dim=0)
    s = (qh @ kh.transpose(-1, -2)) * scale
    s = s.to(H).float()                                   # the kernel stores the scaled scores in fp16 first
    if not softmax:
        return s.to(H)
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.ones(nq, nk, dtype=torch.bool).triu(1), float('-inf'))
    return s.softmax(-1).to(H)


def head_scores(query, key, scale):
    s = (query.float() @ key.float().transpose(-1, -2)) * scale
    return s.to(H).float().softmax(-1).to(H)


def attention_pv(probs, vt, kv_div=1):
    nb, heads, nq, nk = probs.shape
    C = vt.shape[1]
    v = vt[:, :, :nk].float().view(vt.shape[0], heads, C // heads, nk).repeat_interleave(kv_div, dim=0)   # [nb,h,d,nk]
    o = probs.float() @ v.transpose(-1, -2)                                                             # [nb,h,nq,d]
    return o.permute(0, 2, 1, 3).reshape(nb, nq, C).to(H)


def attention(q, k, vt, heads, scale, kv_div=1, nk=None):
    nk = k.shape[1] if nk is None else nk
    qh = _heads(q.contiguous(), heads)
    kh = _heads(k[:, :nk].contiguous(), heads).repeat_interleave(kv_div, dim=0)
    p = ((qh @ kh.transpose(-1, -2)) * scale).softmax(-1)
    C = vt.shape[1]
    v = vt[:, :, :nk].float().view(vt.shape[0], heads, C // heads, nk).repeat_interleave(kv_div, dim=0)
    o = p @ v.transpose(-1, -2)
    return o.permute(0, 2, 1, 3).reshape(q.shape[0], q.shape[1], C).to(H)


def attention_bwd_supported(dh):
    return dh in (40, 64, 80)


def attention_lse(q, k, vt, heads, scale, kv_div=1, nk=None):
    """contract of ops.attention_lse: (out, lse [nb, heads, round_up(nq, 64)] in the log2 domain of the scaled scores)"""
    nk = k.shape[1] if nk is None else nk
    qh = _heads(q.contiguous(), heads)
    kh = _heads(k[:, :nk].contiguous(), heads).repeat_interleave(kv_div, dim=0)
    s = (qh @ kh.transpose(-1, -2)) * scale
    nb, nq = q.shape[:2]
    lse = torch.zeros(nb, heads, (nq + 63) // 64 * 64)
    lse[..., :nq] = torch.logsumexp(s, -1) * 1.4426950408889634
    return attention(q, k, vt, heads, scale, kv_div=kv_div, nk=nk), lse


def attention_bwd(q, k, v, out, dout, lse, heads, scale, kv_div=1, need_kv=True):
    """contract of ops.attention_bwd: P recomputed from lse; dS = scale * P o (dP - delta); fp16 results"""
    nb, nq, C = q.shape
    qh, oh, gh = _heads(q, heads), _heads(out, heads), _heads(dout, heads)
    kh = _heads(k, heads).repeat_interleave(kv_div, dim=0)
    vh = _heads(v, heads).repeat_interleave(kv_div, dim=0)
    p = torch.exp2((qh @ kh.transpose(-1, -2)) * (scale * 1.4426950408889634) - lse[..., :nq, None])
    delta = (gh * oh).sum(-1, keepdim=True)
    ds = (p * (gh @ vh.transpose(-1, -2) - delta) * scale).to(H).float()     # the kernels feed dS / P to the MFMA in fp16
    back = (lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], C).to(H))
    dq = back(ds @ kh)
    if not need_kv:
        return dq, None, None
    assert kv_div == 1
    return dq, back(ds.transpose(-1, -2) @ qh), back(p.to(H).float().transpose(-1, -2) @ gh)


def temporal_attention(q, k, v, B, fq, fk, hw, heads, scale):
    C = q.shape[-1]
    d = C // heads

    def sites(t, f):                                      # [(b f s), C] -> [b, s, h, f, d]
        return t.float().reshape(B, f, hw, heads, d).permute(0, 2, 3, 1, 4)
    p = ((sites(q, fq) @ sites(k, fk).transpose(-1, -2)) * scale).softmax(-1)
    o = p @ sites(v, fk)                                  # [b, s, h, fq, d]
    return o.permute(0, 3, 1, 2, 4).reshape(B * fq * hw, C).to(H)


def group_norm(x, gamma, beta, groups, eps, nimg, silu=False, x2=None, partial_hook=None, count_rows=None):
    xin = x if x2 is None else torch.cat([x, x2], dim=-1)
    C = xin.shape[-1]
    xs = xin.float().reshape(nimg, -1, groups, C // groups)
    rows = xs.shape[1]
    partial = torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], dim=-1)[:, None].contiguous()   # [nimg,1,G,2]
    if partial_hook is not None:
        partial = partial_hook(partial)
    tot = partial.double().sum(1)
    n = (rows if count_rows is None else count_rows) * (C // groups)
    mean = tot[..., 0] / n
    var = (tot[..., 1] / n - mean * mean).clamp_min(0)
    rstd = (var + eps).rsqrt()
    y = (xs - mean[:, None, :, None].float()) * rstd[:, None, :, None].float()
    y = y.reshape(nimg, rows, C) * gamma.float() + beta.float()
    if silu:
        y = F.silu(y)
    return y.reshape(*xin.shape).to(H)


def layer_norm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0, frame_offset=0):
    C = x.shape[-1]
    y = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), eps)
    if pe is not None:
        if pe.shape[0] < frame_offset + frames:
            raise ValueError('positional-encoding table too short')
        rows = y.reshape(-1, C)
        idx = (torch.arange(rows.shape[0]) // rows_per_frame) % frames + frame_offset
        y = (rows + pe.float()[idx]).view(x.shape)
    return y.to(H)


def silu(x):
    return F.silu(x.float()).to(H)


def quick_gelu(x):
    xf = x.float()
    return (xf * torch.sigmoid(1.702 * xf)).to(H)


def axpy(a, b, s=1.0):
    assert a.shape == b.shape
    return (a.float() + s * b.float()).to(H)


def pack_latents(x, cpad=8):
    B, C, Fr, Hh, W = x.shape
    y = torch.zeros(B * Fr, Hh, W, cpad, dtype=H)
    y[..., :C] = x.permute(0, 2, 3, 4, 1).reshape(B * Fr, Hh, W, C)
    return y


def unpack_latents(x, B, cout):
    BF, Hh, W, _ = x.shape
    return x[..., :cout].reshape(B, BF // B, Hh, W, cout).permute(0, 4, 1, 2, 3).contiguous()


def cfg_ddim_step(x, eps_u, eps_c, guidance, alpha_t, alpha_next):
    e = eps_u.float()
    if eps_c is not None:
        e = e + guidance * (eps_c.float() - e)
    x0 = (x.float() - math.sqrt(1.0 - alpha_t) * e) / math.sqrt(alpha_t)
    return (math.sqrt(alpha_next) * x0 + math.sqrt(1.0 - alpha_next) * e).to(H)


def masked_blend(x, src, mask):
    a = src.float()
    return (a + mask.float().reshape(1, *x.shape[1:]) * (x.float() - a)).to(H)


def adapter_scatter(tracks, selected, feat, h, w, rate, out_scale=1.0):
    """vsx_adapter_scatter's contract (adapter_model.py:25-47): fp16 sub-pixel position, fp16 weights, fp16 += in point
    order, finished map times out_scale."""
    Fr, P = tracks.shape[:2]
    C = feat.shape[1]
    out = torch.zeros(Fr, h, w, C, dtype=H)
    r16 = lambda v: torch.tensor(v, dtype=torch.float32).to(H).float().item()     # noqa: E731
    for f in range(Fr):
        for pt in range(P):
            if not int(selected[pt]):
                continue
            px, py = float(tracks[f, pt, 0]), float(tracks[f, pt, 1])
            if px < 0 or py < 0:
                continue
            x, y = r16(r16(px) / rate), r16(r16(py) / rate)
            x1, y1 = int(x), int(y)
            x2, y2 = x1 + 1, y1 + 1
            xf, yf = r16(x - x1), r16(y - y1)
            x1, x2 = max(min(x1, w - 1), 0), max(min(x2, w - 1), 0)
            y1, y2 = max(min(y1, h - 1), 0), max(min(y2, h - 1), 0)
            xm, ym = r16(1.0 - xf), r16(1.0 - yf)
            wgt = [r16(xm * ym), r16(xf * ym), r16(xm * yf), r16(xf * yf)]
            for (xx, yy), wg in zip(((x1, y1), (x2, y1), (x1, y2), (x2, y2)), wgt):
                add = (feat[pt].float() * wg).to(H)
                out[f, yy, xx] = (out[f, yy, xx].float() + add.float()).to(H)
    if out_scale != 1.0:
        out = (out.float() * out_scale).to(H)
    return out


# ---- gradient path (videoswap_amd/autograd.py builds the backward passes from these) ----
def _gelu_parts(g):
    cdf = 0.5 * (1.0 + torch.erf(g * 0.7071067811865476))
    pdf = torch.exp(-0.5 * g * g) * 0.3989422804014327
    return cdf, pdf


def geglu_fwd(y2):
    h, g = y2.float().chunk(2, dim=-1)
    return (h * F.gelu(g)).to(H)


def geglu_bwd(dout, y2):
    h, g = y2.float().chunk(2, dim=-1)
    cdf, pdf = _gelu_parts(g)
    d = dout.float()
    return torch.cat([d * g * cdf, d * h * (cdf + g * pdf)], dim=-1).to(H)


def silu_bwd(dy, x):
    xf = x.float()
    sg = torch.sigmoid(xf)
    return (dy.float() * (sg + xf * sg * (1.0 - sg))).to(H)


def group_norm_bwd(dy, x, gamma, beta, groups, eps, nimg, silu=False, x2=None):
    xin = (x if x2 is None else torch.cat([x, x2], dim=-1)).float().requires_grad_(True)
    C = xin.shape[-1]
    with torch.enable_grad():
        xs = xin.reshape(nimg, -1, groups, C // groups)
        mean = xs.mean((1, 3), keepdim=True)
        var = xs.var((1, 3), unbiased=False, keepdim=True)
        y = ((xs - mean) * (var + eps).rsqrt()).reshape(nimg, -1, C) * gamma.float() + beta.float()
        if silu:
            y = F.silu(y)
        (g,) = torch.autograd.grad(y, xin, dy.float().reshape(y.shape))
    g = g.to(H)
    if x2 is None:
        return g, None
    c1 = x.shape[-1]
    return g[..., :c1].contiguous(), g[..., c1:].contiguous()


def layer_norm_bwd(dy, x, gamma, eps=1e-5):
    xf = x.float().requires_grad_(True)
    with torch.enable_grad():
        y = F.layer_norm(xf, (x.shape[-1],), gamma.float(), None, eps)
        (g,) = torch.autograd.grad(y, xf, dy.float())
    return g.to(H)


def softmax_bwd(probs, dprobs, scale):
    p, dp = probs.float(), dprobs.float()
    ds = scale * p * (dp - (dp * p).sum(-1, keepdim=True))
    dprobs.copy_(ds.to(H))
    return dprobs


def sum_pool2x2(x):
    n, h2, w2, c = x.shape
    return x.float().view(n, h2 // 2, 2, w2 // 2, 2, c).sum((2, 4)).to(H)


def adapter_gather(tracks, selected, dmap, rate, out_scale=1.0):
    Fr, P = tracks.shape[:2]
    _, h, w, C = dmap.shape
    out = torch.zeros(P, C, dtype=torch.float32)
    r16 = lambda v: torch.tensor(v, dtype=torch.float32).to(H).float().item()     # noqa: E731
    for f in range(Fr):
        for pt in range(P):
            if not int(selected[pt]):
                continue
            px, py = float(tracks[f, pt, 0]), float(tracks[f, pt, 1])
            if px < 0 or py < 0:
                continue
            x, y = r16(r16(px) / rate), r16(r16(py) / rate)
            x1, y1 = int(x), int(y)
            x2, y2 = x1 + 1, y1 + 1
            xf, yf = r16(x - x1), r16(y - y1)
            x1, x2 = max(min(x1, w - 1), 0), max(min(x2, w - 1), 0)
            y1, y2 = max(min(y1, h - 1), 0), max(min(y2, h - 1), 0)
            xm, ym = r16(1.0 - xf), r16(1.0 - yf)
            wgt = [r16(xm * ym), r16(xf * ym), r16(xm * yf), r16(xf * yf)]
            for (xx, yy), wg in zip(((x1, y1), (x2, y1), (x1, y2), (x2, y2)), wgt):
                out[pt] += wg * dmap[f, yy, xx].float()
    return (out * out_scale).to(H)


def gemm(desc):
    raise RuntimeError('host emulation: raw vsx_gemm_f16 descriptors are not emulated (call the typed ops)')


def set_option(name, value):
    pass


def prof_pause(paused):
    pass


_NAMES = ['linear', 'linear_vt', 'conv2d', 'attention_scores', 'head_scores', 'attention_pv', 'attention',
          'temporal_attention', 'group_norm', 'layer_norm', 'silu', 'quick_gelu', 'axpy', 'pack_latents',
          'unpack_latents', 'cfg_ddim_step', 'masked_blend', 'adapter_scatter', 'gemm', 'set_option', 'prof_pause',
          'geglu_fwd', 'geglu_bwd', 'silu_bwd', 'group_norm_bwd', 'layer_norm_bwd', 'softmax_bwd', 'sum_pool2x2',
          'adapter_gather', 'attention_lse', 'attention_bwd', 'attention_bwd_supported']


@contextlib.contextmanager
def installed():
    """Swap the kernel table of videoswap_amd.ops (`ops._raw`) for the functions above (and back)."""
    from videoswap_amd import ops
    saved = dict(ops._raw)

    def kernel_like(fn):
        # a real kernel function writes into torch.empty buffers: its result carries no autograd history.  The
        # stand-ins are plain PyTorch, so they run under no_grad — a gradient can then only come from
        # videoswap_amd/autograd.py, exactly as on the GPU
        @functools.wraps(fn)
        def run(*a, **k):
            with torch.no_grad():
                return fn(*a, **k)
        return run
    try:
        for n in _NAMES:
            ops._raw[n] = kernel_like(globals()[n])
        yield ops
    finally:
        ops._raw.clear()
        ops._raw.update(saved)
