"""Gradient path of the denoising kernels: what the adapter training step (trainer_videoswap.py:33-97, train.py:107-224)
needs and nothing more.  The UNet is frozen there, so every op gets a DATA gradient only; the adapter's own two Linear
layers per level also get weight gradients.

`videoswap_amd.ops` routes an op here when autograd is recording and one of its activation arguments requires grad.
Each op is a `torch.autograd.Function` whose forward is the forward kernel function and whose backward is built from
kernel functions again (PyTorch keeps the tape and does layout plumbing — views, transposes, zero-insertion — nothing
else):

    linear / 1x1 conv   dX = dY W        -> the GEMM on a transposed weight copy (cached per parameter version)
    conv 3x3            dX = dY * flip(W)ᵀ -> the implicit-GEMM conv on a flipped / transposed weight copy; stride 2:
                        on the zero-inserted dY; nearest-2x input: followed by a 2x2 sum pool
    GEGLU               forward splits into GEMM + geglu_fwd (the pre-activations are kept), backward geglu_bwd + GEMM
    GroupNorm(+SiLU), LayerNorm, SiLU     dedicated backward kernels
    attention           P is re-materialised per call (attention_scores), dP = dO Vᵀ, dS = softmax_bwd, dQ = dS K,
                        dK = dSᵀ Q, dV = Pᵀ dO: all on the batched GEMM; temporal attention = the same on site-major
                        copies
    adapter_scatter     adapter_gather
"""
import torch

from . import ops

_F16 = torch.float16


def _k(name):
    return ops._raw[name]


# ---- transposed / flipped weight copies, rebuilt when the parameter changes -------------------------------------
_wcache = {}


def _derived(weight, kind, fn):
    key = (id(weight), kind)
    stamp = (weight.data_ptr(), weight._version, weight.dtype, tuple(weight.shape))
    hit = _wcache.get(key)
    if hit is None or hit[0] != stamp or hit[1]() is not weight:
        import weakref
        with torch.no_grad():
            val = fn(weight.detach())
        fresh = hit is None or hit[1]() is not weight
        try:
            ref = weakref.ref(weight)
            if fresh:                           # the entry lives as long as the tensor object it was derived from
                weakref.finalize(weight, _wcache.pop, key, None)
        except TypeError:  # pragma: no cover
            ref = (lambda w=weight: w)
        hit = (stamp, ref, val)
        _wcache[key] = hit
    return hit[2]


def _wt(weight):
    """[N, K] (or 1x1 conv [N, K, 1, 1]) -> [K, N] contiguous: the B operand of dX = dY W"""
    return _derived(weight, 't', lambda w: w.reshape(w.shape[0], -1).t().contiguous())


def _w_dgrad(weight, pad_cout=0):
    """OHWI conv weight [Co, kh, kw, Ci] -> [Ci, kh, kw, Co(+pad)] with the taps flipped: the conv that maps dY to dX"""
    def make(w):
        d = w.permute(3, 1, 2, 0).flip(1, 2)
        if pad_cout and pad_cout != d.shape[-1]:
            d = torch.nn.functional.pad(d, (0, pad_cout - d.shape[-1]))
        return d.contiguous()
    return _derived(weight, ('dgrad', pad_cout), make)


def _need(ctx, i):
    return ctx.needs_input_grad[i]


# ---- GEMM family ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, geglu):
        y2 = None
        if geglu:
            y2 = _k('linear')(x, weight, bias)
            out = _k('geglu_fwd')(y2)
        else:
            out = _k('linear')(x, weight, bias, residual=residual)
        ctx.geglu = geglu
        ctx.xshape = x.shape
        ctx.save_for_backward(weight, y2, x if weight.requires_grad else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        weight, y2, x = ctx.saved_tensors
        dy = dy.contiguous()
        d = _k('geglu_bwd')(dy, y2) if ctx.geglu else dy
        n = d.shape[-1]
        d2 = d.reshape(-1, n)
        dx = dw = db = dres = None
        if _need(ctx, 0):
            dx = _k('linear')(d2, _wt(weight)).view(ctx.xshape)
        if _need(ctx, 1):                       # trainable weight (the adapter's MLPs): dW = dYᵀ X, also on the GEMM
            x2 = x.reshape(-1, x.shape[-1])
            m = d2.shape[0]
            mp = (m + 7) // 8 * 8               # the reduction axis (rows) is padded to the GEMM's 8-element granularity
            dyt = torch.zeros(n, mp, dtype=_F16, device=dy.device)
            dyt[:, :m] = d2.t()
            xt = torch.zeros(x2.shape[1], mp, dtype=_F16, device=dy.device)
            xt[:, :m] = x2.t()
            dw = _k('linear')(dyt, xt).view(weight.shape)
            if _need(ctx, 2):
                ones = torch.zeros(8, mp, dtype=_F16, device=dy.device)
                ones[0, :m] = 1.0
                db = _k('linear')(dyt, ones)[:, 0].contiguous()
        elif _need(ctx, 2):
            raise NotImplementedError('bias gradient without a weight gradient')
        if _need(ctx, 3):
            dres = dy
        return dx, dw, db, dres, None


def linear(x, weight, bias=None, residual=None, geglu=False, out=None, row_stats=False):      # (row_stats: inference-only hint)
    if out is not None:
        raise NotImplementedError('linear(out=...) on the gradient path')
    return _Linear.apply(x, weight, bias, residual, geglu)


class _LinearVT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rows_per_img, ldvt):
        vt = _k('linear_vt')(x, weight, bias, rows_per_img, ldvt)
        ctx.rows = rows_per_img
        ctx.xshape = x.shape
        ctx.save_for_backward(weight)
        return vt

    @staticmethod
    def backward(ctx, dvt):
        (weight,) = ctx.saved_tensors
        n = dvt.shape[1]
        dy = dvt[:, :, :ctx.rows].transpose(1, 2).reshape(-1, n).contiguous()
        return _k('linear')(dy, _wt(weight)).view(ctx.xshape), None, None, None, None


def linear_vt(x, weight, bias, rows_per_img, ldvt=None):
    return _LinearVT.apply(x, weight, bias, rows_per_img, ldvt)


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, residual, weight, bias, rowvec, stride, upsample, rows_per_vec, padding):
        if padding is not None:
            raise NotImplementedError('conv2d with explicit padding on the gradient path (VAE only: frozen)')
        out = _k('conv2d')(x, weight, bias, x2=x2, stride=stride, upsample=upsample, rowvec=rowvec,
                           rows_per_vec=rows_per_vec, residual=residual)
        ctx.stride, ctx.upsample = stride, upsample
        ctx.xshape = x.shape
        ctx.c1 = x.shape[-1]
        ctx.c2 = x2.shape[-1] if x2 is not None else 0
        ctx.save_for_backward(weight)
        return out

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        dy = dy.contiguous()
        n, ho, wo, co = dy.shape
        dx = dx2 = dres = None
        if _need(ctx, 2):
            dres = dy
        if _need(ctx, 0) or _need(ctx, 1):
            cop = (co + 7) // 8 * 8             # conv_out has 4 output channels: the conv kernel wants multiples of 8
            src = dy if cop == co else torch.nn.functional.pad(dy, (0, cop - co))
            if ctx.stride == 2:                 # dX = flip(W) * (dY with zeros inserted between its pixels)
                hin, win = ctx.xshape[1], ctx.xshape[2]
                z = torch.zeros(n, hin, win, cop, dtype=_F16, device=dy.device)
                z[:, ::2, ::2] = src
                src = z
            wd = _w_dgrad(weight, pad_cout=cop if cop != co else 0)
            c1, c2 = ctx.c1, ctx.c2

            def dgrad(lo, hi):
                d = _k('conv2d')(src, wd[lo:hi])
                return _k('sum_pool2x2')(d) if ctx.upsample else d
            if _need(ctx, 0):
                dx = dgrad(0, c1)
            if c2 and _need(ctx, 1):
                dx2 = dgrad(c1, c1 + c2)
        return dx, dx2, dres, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, *, x2=None, stride=1, upsample=False, rowvec=None, rows_per_vec=0, residual=None,
           padding=None):
    return _Conv2d.apply(x, x2, residual, weight, bias, rowvec, stride, upsample, rows_per_vec, padding)


# ---- normalisation / element-wise -------------------------------------------------------------------------------
class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, gamma, beta, groups, eps, nimg, silu, partial_hook, count_rows):
        if partial_hook is not None:
            raise NotImplementedError('frame-sharded GroupNorm on the gradient path')
        y = _k('group_norm')(x, gamma, beta, groups, eps, nimg, silu=silu, x2=x2)
        ctx.cfg = (groups, eps, nimg, silu)
        ctx.save_for_backward(x, x2, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, gamma, beta = ctx.saved_tensors
        groups, eps, nimg, silu = ctx.cfg
        dx, dx2 = _k('group_norm_bwd')(dy.contiguous(), x, gamma, beta, groups, eps, nimg, silu=silu, x2=x2)
        return dx, dx2, None, None, None, None, None, None, None, None


def group_norm(x, gamma, beta, groups, eps, nimg, silu=False, x2=None, partial_hook=None, count_rows=None):
    return _GroupNorm.apply(x, x2, gamma, beta, groups, eps, nimg, silu, partial_hook, count_rows)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pe, rows_per_frame, frames, frame_offset):
        y = _k('layer_norm')(x, gamma, beta, eps, pe=pe, rows_per_frame=rows_per_frame, frames=frames,
                             frame_offset=frame_offset)
        ctx.eps = eps
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        return _k('layer_norm_bwd')(dy.contiguous(), x, gamma, ctx.eps), None, None, None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0, frame_offset=0):
    return _LayerNorm.apply(x, gamma, beta, eps, pe, rows_per_frame, frames, frame_offset)


class _Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _k('silu')(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _k('silu_bwd')(dy.contiguous(), x)


def silu(x):
    return _Silu.apply(x)


class _Axpy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, s):
        ctx.s = s
        return _k('axpy')(a, b, s)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        db = None
        if _need(ctx, 1):
            db = dy if ctx.s == 1.0 else _k('axpy')(dy, dy, ctx.s - 1.0)      # dy + (s - 1) dy
        return (dy if _need(ctx, 0) else None), db, None


def axpy(a, b, s=1.0):
    return _Axpy.apply(a, b, float(s))


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, cout):
        ctx.cs = x.shape[-1]
        return _k('unpack_latents')(x, B, cout)

    @staticmethod
    def backward(ctx, dy):
        return _k('pack_latents')(dy.contiguous(), ctx.cs), None, None


def unpack_latents(x, B, cout):
    return _Unpack.apply(x, B, cout)


# ---- attention ----------------------------------------------------------------------------------------------------
def _pad_last(t, mult=8):
    n = t.shape[-1]
    ld = (n + mult - 1) // mult * mult
    if ld == n:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], ld, dtype=t.dtype, device=t.device)
    out[..., :n] = t
    return out


def _attention_backward(q, k, v, do, heads, scale, kv_div, need_kv):
    """q, do [nb, nq, C]; k, v [nkvb, nk, C] (contiguous) -> (dq, dk, dv); dk / dv only for self-attention."""
    nb, nq, c = q.shape
    nk = k.shape[1]
    probs = _k('attention_scores')(q, k, heads, scale, kv_div=kv_div)                      # [nb, h, nq, nk]
    dprobs = _k('attention_scores')(do, v, heads, 1.0, kv_div=kv_div, softmax=False)       # dP = dO Vᵀ
    ds = _k('softmax_bwd')(probs, dprobs, scale)                                           # over dP
    kt = _pad_last(k.transpose(1, 2))                                                      # [nkvb, C, nk']
    dq = _k('attention_pv')(ds, kt, kv_div=kv_div)                                         # dS K
    dk = dv = None
    if need_kv:
        if kv_div != 1:
            raise NotImplementedError('key / value gradients of a shared (text) context')
        ds_t = ds.transpose(2, 3).contiguous()                                             # [nb, h, nk, nq]
        dk = _k('attention_pv')(ds_t, _pad_last(q.transpose(1, 2)))                        # dSᵀ Q
        p_t = probs.transpose(2, 3).contiguous()
        dv = _k('attention_pv')(p_t, _pad_last(do.transpose(1, 2)))                        # Pᵀ dO
    return dq, dk, dv


class _Attention(torch.autograd.Function):
    """Fused attention.  For the head dims the flash backward exists for (40 / 64 / 80: every spatial attention of the
    64x64 and 32x32 levels) the forward also keeps the row log-sum-exp and the output, and the backward recomputes the
    probabilities tile by tile (csrc/attention_bwd.hip); the other head dims (d = 160: 256 keys) keep the materialised
    path below."""

    @staticmethod
    def forward(ctx, q, k, vt, heads, scale, kv_div, nk):
        nk_eff = k.shape[1] if nk is None else nk
        flash = 'attention_lse' in ops._raw and _k('attention_bwd_supported')(q.shape[-1] // heads)
        ctx.cfg = (heads, scale, kv_div, nk_eff, flash)
        if flash:
            out, lse = _k('attention_lse')(q, k, vt, heads, scale, kv_div=kv_div, nk=nk)
            ctx.save_for_backward(q, k, vt, out, lse)
            return out
        out = _k('attention')(q, k, vt, heads, scale, kv_div=kv_div, nk=nk)
        ctx.save_for_backward(q, k, vt)
        return out

    @staticmethod
    def backward(ctx, do):
        heads, scale, kv_div, nk, flash = ctx.cfg
        need_kv = _need(ctx, 1) or _need(ctx, 2)
        q, k, vt = ctx.saved_tensors[:3]
        kc = k[:, :nk].contiguous()
        v = vt[:, :, :nk].transpose(1, 2).contiguous()
        if flash:
            out, lse = ctx.saved_tensors[3:]
            dq, dk, dv = _k('attention_bwd')(q.contiguous(), kc, v, out, do.contiguous(), lse, heads, scale, kv_div=kv_div,
                                             need_kv=need_kv)
        else:
            dq, dk, dv = _attention_backward(q.contiguous(), kc, v, do.contiguous(), heads, scale, kv_div, need_kv)
        dk_full = dvt = None
        if need_kv:
            dk_full = dk if nk == k.shape[1] else torch.nn.functional.pad(dk, (0, 0, 0, k.shape[1] - nk))
            dvt = torch.zeros_like(vt)
            dvt[:, :, :nk] = dv.transpose(1, 2)
        return dq, dk_full, dvt, None, None, None, None


def attention(q, k, vt, heads, scale, kv_div=1, nk=None):
    return _Attention.apply(q, k, vt, heads, scale, kv_div, nk)


class _TemporalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, B, fq, fk, hw, heads, scale):
        out = _k('temporal_attention')(q, k, v, B, fq, fk, hw, heads, scale)
        ctx.cfg = (B, fq, fk, hw, heads, scale)
        ctx.save_for_backward(q, k, v)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        B, fq, fk, hw, heads, scale = ctx.cfg
        c = q.shape[-1]

        def sites(t, f):                        # [(b f s), C] -> [(b s), f, C]
            return t.reshape(B, f, hw, c).permute(0, 2, 1, 3).reshape(B * hw, f, c).contiguous()

        def frames(t, f):                       # back
            return t.reshape(B, hw, f, c).permute(0, 2, 1, 3).reshape(B * f * hw, c).contiguous()
        qs, ks, vs, dos = sites(q, fq), sites(k, fk), sites(v, fk), sites(do, fq)
        step = max(1, 65535 // heads)           # batched-GEMM grid limit: images x heads per launch
        dq, dk, dv = [], [], []
        for i in range(0, B * hw, step):
            a, b_, c_ = _attention_backward(qs[i:i + step], ks[i:i + step], vs[i:i + step], dos[i:i + step], heads,
                                            scale, 1, True)
            dq.append(a); dk.append(b_); dv.append(c_)
        cat = (lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts))
        return frames(cat(dq), fq), frames(cat(dk), fk), frames(cat(dv), fk), None, None, None, None, None, None


def temporal_attention(q, k, v, B, fq, fk, hw, heads, scale):
    return _TemporalAttention.apply(q, k, v, B, fq, fk, hw, heads, scale)


# ---- adapter ----------------------------------------------------------------------------------------------------
class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tracks, selected, feat, h, w, rate, out_scale):
        ctx.cfg = (rate, out_scale)
        ctx.save_for_backward(tracks, selected)
        return _k('adapter_scatter')(tracks, selected, feat, h, w, rate, out_scale)

    @staticmethod
    def backward(ctx, dmap):
        tracks, selected = ctx.saved_tensors
        rate, out_scale = ctx.cfg
        return None, None, _k('adapter_gather')(tracks, selected, dmap.contiguous(), rate, out_scale), None, None, None, None


def adapter_scatter(tracks, selected, feat, h, w, rate, out_scale=1.0):
    return _Scatter.apply(tracks, selected, feat, h, w, rate, out_scale)


_TABLE = {'linear': linear, 'linear_vt': linear_vt, 'conv2d': conv2d, 'group_norm': group_norm,
          'layer_norm': layer_norm, 'silu': silu, 'axpy': axpy, 'unpack_latents': unpack_latents,
          'attention': attention, 'temporal_attention': temporal_attention, 'adapter_scatter': adapter_scatter}


def dispatch(name, *args, **kwargs):
    fn = _TABLE.get(name)
    if fn is None:
        raise NotImplementedError(f'videoswap_amd.ops.{name} has no gradient: the training step uses the fused '
                                  f'attention processors and never differentiates this op')
    return fn(*args, **kwargs)
