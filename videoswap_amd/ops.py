"""Tensor-level wrappers over the libvsx C ABI.

PyTorch is used for device memory and the HIP stream only: every function here takes fp16 CUDA
(ROCm) tensors, hands raw pointers + the current stream to a hand-written gfx950 kernel and
returns the output tensor.  There is no fallback implementation.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import GemmDesc, check

_F16 = torch.float16


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream (the raw-handle query is ~10x cheaper than building a Stream object;
    with ~700 launches per UNet call the difference is several ms of host time per forward)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, name, dtype=_F16):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.VsxError(f'{name}: expected a GPU tensor (the denoising path has no CPU implementation)')
    if t.dtype != dtype:
        raise _lib.VsxError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.VsxError(f'{name}: expected a contiguous tensor, got strides {t.stride()}')


def round_up(x, m):
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------------
# GEMM family
# --------------------------------------------------------------------------------------------
class FlopCounter:
    """Algorithmic FLOP (2 x MAC, true dimensions, no padding) of the MFMA kernels launched while enabled.  `gemm` counts what
    the launches multiply; `gemm_saved` what the reference's formulation would have multiplied on top of that (the five taps
    per output pixel that the sub-pixel form of the nearest-2x convolutions never computes)."""
    enabled = False
    gemm = 0.0
    gemm_saved = 0.0
    attention = 0.0

    @classmethod
    def reset(cls, enabled=True):
        cls.enabled, cls.gemm, cls.gemm_saved, cls.attention = enabled, 0.0, 0.0, 0.0


_split_ws = {}      # (device, stream) -> grow-only fp32 scratch for split-K partial sums (stream-ordered reuse)


# VSX_GEMM_LOG=<file>: one line per vsx_gemm_f16 launch (shape, loader mode, epilogue), in launch order — joined by
# tools/pmc_by_shape.py with rocprofv3's per-dispatch counters (the i-th GEMM dispatch is the i-th line).  Measurement only.
_gemm_log = None
if os.environ.get('VSX_GEMM_LOG'):
    _gemm_log = open(os.environ['VSX_GEMM_LOG'], 'w')


def gemm(desc, k_flop=None):
    """k_flop: the K the launch actually multiplies when that is not desc.K (sub-pixel form of the nearest-2x convolution)"""
    if _gemm_log is not None:
        _gemm_log.write(f'{desc.M} {desc.N} {desc.K} {desc.batch0 * desc.batch1} {desc.a_mode} {desc.ks} {desc.stride} '
                        f'{desc.upsample} {desc.C1} {desc.C2} {desc.H} {desc.W} {desc.geglu} {desc.c_mode} '
                        f'{1 if desc.residual else 0} {1 if desc.rowvec else 0} {1 if desc.bias else 0}\n')
        _gemm_log.flush()
    if FlopCounter.enabled:
        cols = desc.N * (2 if desc.geglu else 1)
        FlopCounter.gemm += 2.0 * desc.M * cols * (desc.K if k_flop is None else k_flop) * desc.batch0 * desc.batch1
        if k_flop is not None:
            FlopCounter.gemm_saved += 2.0 * desc.M * cols * (desc.K - k_flop) * desc.batch0 * desc.batch1
    lib = _lib.load()
    need = lib.vsx_gemm_workspace(ctypes.byref(desc)) if (desc.M <= 20480 and desc.K >= 768) else 0
    stream = _stream()
    if need > 0:
        # one scratch buffer per (device, stream): launches on one stream are ordered, two streams (or a graph capture
        # next to eager work) must not share partial sums
        key = (torch.cuda.current_device(), stream.value)
        ws = _split_ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=f'cuda:{key[0]}')
            _split_ws[key] = ws
        desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    check(lib.vsx_gemm_f16(ctypes.byref(desc), stream), 'vsx_gemm_f16')


# --------------------------------------------------------------------------------------------
# LayerNorm folded into the Linear that consumes it (include/vsx.h: vsx_gemm_desc.rowscale / colvec)
# --------------------------------------------------------------------------------------------
LN_FUSE = os.environ.get('VSX_LN_FUSE', '1') != '0'     # 0: every LayerNorm runs as its own kernel (A/B runs)


# VSX_ROW_STATS_PRODUCER=0: LayerNorm row statistics always by the standalone pass (A/B runs)
ROW_STATS_FROM_PRODUCER = os.environ.get('VSX_ROW_STATS_PRODUCER', '1') != '0'
ROW_STATS_MAX_PARTS = 64           # vsx_row_stats_combine's limit (norm.hip): N <= 3200 on the persistent kernel's 6 parts per 320 columns


class DeferredLN:
    """A LayerNorm that has not been applied: `x` is the RAW activation, and the Linear(s) consuming the normalised tensor
    (attention.py:182,199,205 norm1/2/3 -> to_q / to_k / to_v / GEGLU; motion_module.py:213,219) apply it inside their
    GEMM through  LN(x) W^T + b = rstd * (x (W o gamma)^T) - rstd * mean * sum_k (W o gamma) + (beta W^T + b)  (+ pe W^T for
    the temporal positional encoding).  The normalised tensor is never written; the row statistics are computed once per
    DeferredLN (`stats()`), whatever the number of consumers.  Quacks like the tensor the processors expect as far as they
    use it (shape, reshape / view, dtype, device); `materialize()` is the ordinary kernel for anything else."""
    is_cuda = True
    requires_grad = False

    def __init__(self, x, gamma, beta, eps, pe=None, rows_per_frame=0, frames=0, frame_offset=0, _shared=None):
        self.x, self.gamma, self.beta, self.eps = x, gamma, beta, eps
        self.pe, self.rows_per_frame, self.frames, self.frame_offset = pe, rows_per_frame, frames, frame_offset
        self._shared = _shared if _shared is not None else {}          # stats / materialised tensor, shared by reshapes
        # partial row sums the producing GEMM hung on its output (they describe the ROWS: reshapes keep them)
        if 'row_parts' not in self._shared:
            self._shared['row_parts'] = getattr(x, '_vsx_rowparts', None)

    row_parts = property(lambda self: self._shared.get('row_parts'))

    shape = property(lambda self: self.x.shape)
    dtype = property(lambda self: self.x.dtype)
    device = property(lambda self: self.x.device)

    def dim(self):
        return self.x.dim()

    def numel(self):
        return self.x.numel()

    def contiguous(self):
        return self

    def _like(self, x):
        return DeferredLN(x, self.gamma, self.beta, self.eps, self.pe, self.rows_per_frame, self.frames, self.frame_offset,
                          self._shared)

    def reshape(self, *shape):
        return self._like(self.x.reshape(*shape))

    def view(self, *shape):
        return self._like(self.x.view(*shape))

    def stats(self):
        """[M, 2] fp32 = (rstd, -rstd * mean) per row, computed once: from the partial sums the GEMM that produced x
        wrote in its epilogue (`linear(..., row_stats=True)` hangs them on its output tensor) where there are any,
        otherwise by a pass over x (vsx_row_stats)"""
        st = self._shared.get('stats')
        if st is None:
            C = self.x.shape[-1]
            M = self.x.numel() // C
            st = torch.empty(M, 2, dtype=torch.float32, device=self.x.device)
            parts = self.row_parts
            if parts is not None and parts.shape[0] == M and ROW_STATS_FROM_PRODUCER:
                check(_lib.load().vsx_row_stats_combine(_p(parts), M, parts.shape[1], C, float(self.eps), _p(st), _stream()),
                      'vsx_row_stats_combine')
            else:
                check(_lib.load().vsx_row_stats(_p(self.x), M, C, float(self.eps), _p(st), _stream()), 'vsx_row_stats')
            self._shared['stats'] = st
        return st

    def materialize(self):
        y = self._shared.get('y')
        if y is None:
            y = _raw['layer_norm'](self.x.reshape(-1, self.x.shape[-1]), self.gamma, self.beta, self.eps, pe=self.pe,
                                   rows_per_frame=self.rows_per_frame, frames=self.frames, frame_offset=self.frame_offset)
            self._shared['y'] = y
        return y.view(self.x.shape)


_fold_cache = {}      # (id(weight), id(gamma)) -> (stamp, weight ref, W o gamma, c1, c2, {pe products}, gamma ref)


def fold_cache_tensors():
    """Every folded operand alive right now.  A captured HIP graph bakes their addresses into its kernel arguments:
    graphs.GraphCache keeps this list with the graph entry, so the memory outlives any eviction here."""
    out = []
    for hit in _fold_cache.values():
        out.extend((hit[2], hit[3], hit[4]))
        out.extend(hit[5].values())
    out.extend(hit[2] for hit in _subpixel_cache.values())
    return out


def _ln_folded(weight, bias, ln):
    """W' = W o gamma (fp16, the B operand), c1[n] = sum_k W'[n, k] (fp32), c2 = beta W^T + b (fp16, the epilogue bias);
    rebuilt when any of the parameters changes (LoRA merge / load_state_dict bump the version).  An entry lives exactly as
    long as its weight tensor (a finalizer on the weight evicts it: the fused-projection tensors that a LoRA merge
    rebuilds do not leave stale folded copies behind, and nothing is ever dropped while its weight — and therefore a
    graph captured over it — can still be used)."""
    import weakref
    key = (id(weight), id(ln.gamma))
    stamp = tuple((t.data_ptr(), t._version) for t in (weight, ln.gamma, ln.beta) + ((bias,) if bias is not None else ()))
    hit = _fold_cache.get(key)
    if hit is None or hit[0] != stamp or hit[1]() is not weight or hit[6]() is not ln.gamma:
        with torch.no_grad():
            w32 = weight.detach().reshape(weight.shape[0], -1).float()
            wf = (w32 * ln.gamma.detach().float()[None, :]).to(_F16).contiguous()
            c1 = wf.float().sum(1).contiguous()
            c2 = w32 @ ln.beta.detach().float()
            if bias is not None:
                c2 = c2 + bias.detach().float()
            c2 = c2.to(_F16).contiguous()
        fresh = hit is None or hit[1]() is not weight
        hit = (stamp, weakref.ref(weight), wf, c1, c2, {}, weakref.ref(ln.gamma))
        _fold_cache[key] = hit
        if fresh:
            weakref.finalize(weight, _fold_cache.pop, key, None)
    return hit


def _ln_pe_rows(hit, weight, ln, M):
    """pe W^T as the GEMM's row-vector term: [M / rows_per_frame, N] (row m takes frame (m / rows_per_frame) % frames)"""
    if ln.pe is None:
        return None, 0
    nblk = M // ln.rows_per_frame
    key = (ln.pe.data_ptr(), ln.pe._version, ln.frame_offset, ln.frames, nblk)
    rv = hit[5].get(key)
    if rv is None:
        with torch.no_grad():
            pe = ln.pe[ln.frame_offset:ln.frame_offset + ln.frames].float()
            rows = (pe @ weight.detach().reshape(weight.shape[0], -1).float().t()).to(_F16)        # [frames, N]
            rv = rows.repeat(nblk // ln.frames, 1).contiguous()
        if len(hit[5]) > 16:
            hit[5].clear()
        hit[5][key] = rv
    return rv, ln.rows_per_frame


def linear(x, weight, bias=None, residual=None, geglu=False, out=None, row_stats=False):
    """y = x @ weight.T (+bias) (+residual); geglu: weight is [2N,K], y = h * gelu(g).

    x [..., K] -> [..., N].  Replaces nn.Linear / 1x1 conv / diffusers GEGLU.  `x` may be a DeferredLN: the LayerNorm is
    then applied inside the GEMM.  `row_stats`: a LayerNorm follows — where the launch can (vsx_gemm_rowstats_parts), its
    epilogue also writes per-row partial sums of the output, which then travel with the returned tensor
    (`y._vsx_rowparts`, picked up by layers.LayerNorm / DeferredLN.stats); a hint, never an obligation.
    """
    ln = x if isinstance(x, DeferredLN) else None
    if ln is not None:
        x = ln.x
    _chk(x, 'x'); _chk(weight, 'weight'); _chk(bias, 'bias'); _chk(residual, 'residual')
    K = x.shape[-1]
    M = x.numel() // K
    if weight.dim() == 4:  # 1x1 conv weight [N,K,1,1]
        weight = weight.reshape(weight.shape[0], weight.shape[1])
    n_rows = weight.shape[0]
    if weight.shape[1] != K:
        raise _lib.VsxError(f'linear: weight {tuple(weight.shape)} does not match input features {K}')
    N = n_rows // 2 if geglu else n_rows
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=_F16, device=x.device)
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch0 = d.batch1 = 1
    d.A = x.data_ptr(); d.lda = K
    d.B = weight.data_ptr(); d.ldb = K
    d.C = out.data_ptr(); d.ldc = N
    d.bias = bias.data_ptr() if bias is not None else None
    keep = None
    if ln is not None:
        hit = _ln_folded(weight, bias, ln)
        rv, rpv = _ln_pe_rows(hit, weight, ln, M)
        if rv is not None and geglu:
            raise _lib.VsxError('linear: a positional encoding cannot be folded into a GEGLU projection')
        st = ln.stats()
        keep = (hit, rv, st)
        d.B = hit[2].data_ptr()
        d.bias = hit[4].data_ptr()
        d.rowscale, d.colvec = st.data_ptr(), hit[3].data_ptr()
        if rv is not None:
            d.rowvec, d.rows_per_vec = rv.data_ptr(), rpv
    if residual is not None:
        if residual.numel() != M * N:
            raise _lib.VsxError('linear: residual shape mismatch')
        d.residual = residual.data_ptr(); d.ldr = N
    d.geglu = 1 if geglu else 0
    d.alpha = 1.0
    parts = None
    if row_stats and ln is None and not geglu and ROW_STATS_FROM_PRODUCER:
        nparts = int(_lib.load().vsx_gemm_rowstats_parts(ctypes.byref(d)))
        if 0 < nparts <= ROW_STATS_MAX_PARTS:      # (a hint, never an obligation: wider outputs take the stand-alone statistics pass)
            parts = torch.empty(M, nparts, 2, dtype=torch.float32, device=x.device)
            d.rowstats, d.rowstats_parts = parts.data_ptr(), nparts
    gemm(d)
    del keep
    if parts is not None:
        out._vsx_rowparts = parts
    return out


def linear_vt(x, weight, bias, rows_per_img, ldvt=None):
    """V^T projection: x [nimg*rows, K] -> VT [nimg, N, ldvt] with VT[i, n, r] = (x @ W.T)[i*rows + r, n].

    Feeds vsx_attention_f16 / attention_pv (V is consumed key-contiguous).  Columns >= rows of VT are left
    untouched (the consumers mask them).
    """
    ln = x if isinstance(x, DeferredLN) else None
    if ln is not None:
        x = ln.x
    _chk(x, 'x'); _chk(weight, 'weight'); _chk(bias, 'bias')
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    nimg = M // rows_per_img
    if ldvt is None:
        ldvt = round_up(rows_per_img, 8)
    vt = torch.empty(nimg, N, ldvt, dtype=_F16, device=x.device)
    if ldvt != rows_per_img:
        vt.zero_()
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch0 = d.batch1 = 1
    d.A = x.data_ptr(); d.lda = K
    d.B = weight.data_ptr(); d.ldb = K
    d.C = vt.data_ptr(); d.ldc = ldvt
    d.c_mode = 1; d.c_rows_per_img = rows_per_img; d.c_img_stride = N * ldvt
    d.bias = bias.data_ptr() if bias is not None else None
    keep = None
    if ln is not None:
        if ln.pe is not None:
            raise _lib.VsxError('linear_vt: a positional encoding cannot be folded into the transposed V projection')
        hit = _ln_folded(weight, bias, ln)
        st = ln.stats()
        keep = (hit, st)
        d.B = hit[2].data_ptr()
        d.bias = hit[4].data_ptr()
        d.rowscale, d.colvec = st.data_ptr(), hit[3].data_ptr()
    d.alpha = 1.0
    gemm(d)
    del keep
    return vt


# VSX_CONV_SUBPIXEL=0: the nearest-2x convolutions always as nine taps on the upsampled image (A/B runs)
CONV_SUBPIXEL = os.environ.get('VSX_CONV_SUBPIXEL', '1') != '0'
_subpixel_cache = {}      # id(weight) -> (stamp, weight ref, [4, Cout, 3, 3, C] fp16)


def subpixel_weights(weight):
    """Sub-pixel form of `conv3x3(nearest_2x(x))` (Upsample3D, resnet.py:54,66): a 3x3 window on the upsampled image meets
    only 2 x 2 source pixels, so output pixel (2 i + ph, 2 j + pw) is a 2 x 2-tap convolution of the SOURCE with the filter
    rows / columns that land on the same source pixel added up (fp32 sums, rounded to fp16 once):
        ph = 0: source rows (i - 1, i) <- filter rows ({0}, {1, 2});   ph = 1: source rows (i, i + 1) <- ({0, 1}, {2})
    and the same along the columns.  Returns [4, Cout, 3, 3, C] (class = 2 ph + pw): the four taps of class (ph, pw) sit at
    window positions (ph.., pw..) of a plain pad-1 3x3 window on the source, the other five are zero and never read
    (include/vsx.h: vsx_gemm_desc.upsample = 2).  4 / 9 of the multiplications of the nine-tap form.  Cached per weight."""
    import weakref
    key = id(weight)
    stamp = (weight.data_ptr(), weight._version)
    hit = _subpixel_cache.get(key)
    if hit is None or hit[0] != stamp or hit[1]() is not weight:
        with torch.no_grad():
            w = weight.detach().float()                                   # [Cout, 3, 3, C]
            rows = {0: ((0, (0,)), (1, (1, 2))), 1: ((1, (0, 1)), (2, (2,)))}      # parity -> ((window tap, filter taps), ...)
            out = torch.zeros(4, *w.shape, dtype=torch.float32, device=w.device)
            for ph in (0, 1):
                for pw in (0, 1):
                    for th, fh in rows[ph]:
                        for tw, fw in rows[pw]:
                            out[2 * ph + pw, :, th, tw] = sum(w[:, a, b] for a in fh for b in fw)
            out = out.to(_F16).contiguous()
        fresh = hit is None or hit[1]() is not weight
        hit = (stamp, weakref.ref(weight), out)
        _subpixel_cache[key] = hit
        if fresh:
            weakref.finalize(weight, _subpixel_cache.pop, key, None)
    return hit[2]


def _subpixel_eligible(nimg, Hs, Ws, C1, Cout, ks, stride, x2, rowvec, residual, padding):
    """Mirrors the library's own conditions (gemm.hip: upsample == 2): bias-only single-source 3x3, and a launch the
    persistent kernel's 256-row tiles take (enough tiles, whole tiles per class)."""
    if not CONV_SUBPIXEL or ks != 3 or stride != 1 or x2 is not None or rowvec is not None or residual is not None \
            or padding is not None or C1 % 64 or Cout % 320:
        return False
    if _options.get('gemm_pp', 1) not in (1, 2) or _options.get('tile_tune', 0) != 0:
        return False                    # the persistent kernel is switched off / a tile or a K split is forced (A/B runs, tests)
    mc = nimg * Hs * Ws
    return mc % 256 == 0 and (4 * mc // 256) * (Cout // 320) >= 192


def conv2d(x, weight, bias=None, *, x2=None, stride=1, upsample=False, rowvec=None, rows_per_vec=0,
           residual=None, padding=None):
    """Channels-last conv as implicit GEMM.

    x [nimg, H, W, C1] (+ x2 [nimg, H, W, C2] concatenated on C); weight [Cout, ks, ks, C1+C2] contiguous
    (== a channels_last-format nn.Conv2d weight); returns [nimg, Ho, Wo, Cout].  `upsample`: x/x2 are at half
    resolution and are read as their nearest-2x upsampling.  rowvec [nvec, Cout] is added to rows
    [i*rows_per_vec, (i+1)*rows_per_vec) (time embedding); residual [nimg, Ho, Wo, Cout] is added last.
    `padding` = (before, after) zero rows/columns (default ks//2 each); (0, 1) is diffusers' Downsample2D(padding=0).
    """
    _chk(x, 'x'); _chk(x2, 'x2'); _chk(weight, 'weight'); _chk(bias, 'bias'); _chk(rowvec, 'rowvec')
    _chk(residual, 'residual')
    nimg, Hs, Ws, C1 = x.shape
    C2 = x2.shape[-1] if x2 is not None else 0
    Cout, ks = weight.shape[0], weight.shape[1]
    if weight.shape[2] != ks or weight.shape[3] != C1 + C2:
        raise _lib.VsxError(f'conv2d: weight {tuple(weight.shape)} does not match input channels {C1}+{C2}')
    H, W = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
    pad_lo, pad_hi = (ks // 2, ks // 2) if padding is None else padding      # (before, after) on both spatial axes
    Ho = (H + pad_lo + pad_hi - ks) // stride + 1
    Wo = (W + pad_lo + pad_hi - ks) // stride + 1
    out = torch.empty(nimg, Ho, Wo, Cout, dtype=_F16, device=x.device)
    d = GemmDesc()
    if padding is not None:
        d.pad_lo, d.pad_hi = pad_lo, pad_hi
    d.M, d.N, d.K = nimg * Ho * Wo, Cout, ks * ks * (C1 + C2)
    d.batch0 = d.batch1 = 1
    d.A = x.data_ptr(); d.A2 = x2.data_ptr() if x2 is not None else None
    d.a_mode = 1; d.H, d.W, d.C1, d.C2 = H, W, C1, C2
    d.ks, d.stride, d.upsample = ks, stride, 1 if upsample else 0
    d.B = weight.data_ptr(); d.ldb = d.K
    d.C = out.data_ptr(); d.ldc = Cout
    d.bias = bias.data_ptr() if bias is not None else None
    if upsample and _subpixel_eligible(nimg, Hs, Ws, C1, Cout, ks, stride, x2, rowvec, residual, padding):
        w4 = subpixel_weights(weight)
        d.upsample = 2
        d.B = w4.data_ptr()
        d.alpha = 1.0
        gemm(d, k_flop=4 * C1)
        return out
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.rows_per_vec = rows_per_vec
    if residual is not None:
        if residual.numel() != out.numel():
            raise _lib.VsxError('conv2d: residual shape mismatch')
        d.residual = residual.data_ptr(); d.ldr = Cout
    d.alpha = 1.0
    gemm(d)
    return out


def attention_scores(q, k, heads, scale, kv_div=1, causal=False, softmax=True):
    """probs = softmax(scale * q k^T) materialised (the Prompt-to-Prompt hook path).

    q [nb, nq, C], k [nkvb, nk, C] -> probs [nb, heads, nq, nk] (a view of a buffer whose rows are padded to a
    multiple of 8 so that probs @ V can stream it with 16-byte loads)."""
    _chk(q, 'q'); _chk(k, 'k')
    nb, nq, C = q.shape
    nk = k.shape[1]
    dh = C // heads
    ld = round_up(nk, 8)
    buf = torch.empty(nb, heads, nq, ld, dtype=_F16, device=q.device)
    if ld != nk:
        buf.zero_()
    if kv_div == 1:
        d = GemmDesc()
        d.M, d.N, d.K = nq, nk, dh
        d.batch0, d.batch1 = nb, heads
        d.A = q.data_ptr(); d.lda = C; d.a_bs0 = nq * C; d.a_bs1 = dh
        d.B = k.data_ptr(); d.ldb = C; d.b_bs0 = nk * C; d.b_bs1 = dh
        d.C = buf.data_ptr(); d.ldc = ld; d.c_bs0 = heads * nq * ld; d.c_bs1 = nq * ld
        d.alpha = float(scale)
        gemm(d)
    else:
        for kb in range(k.shape[0]):  # text K shared by kv_div consecutive images
            d = GemmDesc()
            d.M, d.N, d.K = nq, nk, dh
            d.batch0, d.batch1 = kv_div, heads
            sl = q[kb * kv_div:(kb + 1) * kv_div]
            d.A = sl.data_ptr(); d.lda = C; d.a_bs0 = nq * C; d.a_bs1 = dh
            d.B = k[kb].data_ptr(); d.ldb = C; d.b_bs0 = 0; d.b_bs1 = dh
            d.C = buf[kb * kv_div:(kb + 1) * kv_div].data_ptr(); d.ldc = ld
            d.c_bs0 = heads * nq * ld; d.c_bs1 = nq * ld
            d.alpha = float(scale)
            gemm(d)
    if not softmax:                  # the raw scaled products (gradient path: dP = dO V^T has the same shape)
        return buf[..., :nk]
    if causal:
        check(_lib.load().vsx_softmax_rows_causal(_p(buf), nb * heads * nq, nk, ld, nq, _stream()),
              'vsx_softmax_rows_causal')
    else:
        check(_lib.load().vsx_softmax_rows(_p(buf), nb * heads * nq, nk, ld, _stream()), 'vsx_softmax_rows')
    return buf[..., :nk]


def head_scores(query, key, scale):
    """softmax(scale * q k^T) for head-batched operands [B*heads, N, d] x [B*heads, Nk, d] -> [B*heads, N, Nk]
    (diffusers `Attention.get_attention_scores`, the surface foreign processors call)."""
    query, key = query.contiguous(), key.contiguous()
    _chk(query, 'query'); _chk(key, 'key')
    nbh, nq, d = query.shape
    nk = key.shape[1]
    probs = torch.empty(nbh, nq, nk, dtype=_F16, device=query.device)
    g = GemmDesc()
    g.M, g.N, g.K = nq, nk, d
    g.batch0, g.batch1 = nbh, 1
    g.A = query.data_ptr(); g.lda = d; g.a_bs0 = nq * d
    g.B = key.data_ptr(); g.ldb = d; g.b_bs0 = nk * d
    g.C = probs.data_ptr(); g.ldc = nk; g.c_bs0 = nq * nk
    g.alpha = float(scale)
    gemm(g)
    check(_lib.load().vsx_softmax_rows(_p(probs), nbh * nq, nk, nk, _stream()), 'vsx_softmax_rows')
    return probs


def attention_pv(probs, vt, kv_div=1):
    """out[b, q, h*d + c] = sum_t probs[b, h, q, t] * vt[b // kv_div, h*d + c, t]; probs from attention_scores
    (possibly edited in place by a controller; must still be a view of a buffer with padded rows)."""
    nb, heads, nq, nk = probs.shape
    ld = probs.stride(2)
    if probs.stride(3) != 1 or probs.stride(1) != nq * ld or probs.stride(0) != heads * nq * ld or ld % 8:
        ld = round_up(nk, 8)
        buf = torch.zeros(nb, heads, nq, ld, dtype=_F16, device=probs.device)
        buf[..., :nk] = probs
        probs = buf[..., :nk]
    if probs.dtype != _F16 or not probs.is_cuda:
        raise _lib.VsxError('attention_pv: probs must be an fp16 GPU tensor')
    _chk(vt, 'vt')
    C, ldvt = vt.shape[1], vt.shape[2]
    dh = C // heads
    out = torch.empty(nb, nq, C, dtype=_F16, device=probs.device)
    # K (reduction) = nk rounded up to 8: padded columns of probs are zero, padded VT columns masked by them
    kred = round_up(nk, 8)
    if ldvt < kred:
        raise _lib.VsxError('attention_pv: vt rows are shorter than round_up(nk, 8)')
    groups = [(0, nb, 0)] if kv_div == 1 else [(kb * kv_div, kv_div, kb) for kb in range(vt.shape[0])]
    for (b0, cnt, kb) in groups:
        d = GemmDesc()
        d.M, d.N, d.K = nq, dh, kred
        d.batch0, d.batch1 = cnt, heads
        d.A = probs[b0:].data_ptr(); d.lda = ld; d.a_bs0 = heads * nq * ld; d.a_bs1 = nq * ld
        d.B = vt[kb if kv_div != 1 else b0:].data_ptr(); d.ldb = ldvt
        d.b_bs0 = 0 if kv_div != 1 else C * ldvt; d.b_bs1 = dh * ldvt
        d.C = out[b0:].data_ptr(); d.ldc = C; d.c_bs0 = nq * C; d.c_bs1 = dh
        d.alpha = 1.0
        gemm(d)
    return out


def _rows_view(t, name):
    """[n, rows, C] tensor or column-slice view of a wider [n, rows, W] buffer (fused q|k|v projections): returns
    (row stride, batch stride) in elements after checking what the kernels need."""
    _chk_view(t, name)
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) % 8 or t.stride(0) % 8 or (t.data_ptr() % 16):
        raise _lib.VsxError(f'{name}: need unit inner stride, 16-byte aligned base, strides multiple of 8 (got {t.stride()})')
    return t.stride(1), t.stride(0)


def _chk_view(t, name):
    if not t.is_cuda or t.dtype != _F16:
        raise _lib.VsxError(f'{name}: expected an fp16 GPU tensor')


def attention(q, k, vt, heads, scale, kv_div=1, nk=None):
    """Fused softmax(q k^T * scale) v.  q [nb,nq,C]; k [nkvb,nk,C] (both may be column slices of a fused projection
    buffer); vt [nkvb,C,ldvt] -> [nb,nq,C]."""
    _chk(vt, 'vt')
    ldq, q_bs = _rows_view(q, 'q')
    ldk, k_bs = _rows_view(k, 'k')
    nb, nq, C = q.shape
    nk = k.shape[1] if nk is None else nk
    dh = C // heads
    out = torch.empty(nb, nq, C, dtype=_F16, device=q.device)
    if FlopCounter.enabled:
        FlopCounter.attention += 4.0 * nb * heads * nq * nk * dh
    check(_lib.load().vsx_attention_f16(_p(q), _p(k), _p(vt), _p(out), nb, heads, nq, nk, dh, ldq, ldk, vt.shape[2], C,
                                        q_bs, k_bs, C * vt.shape[2], nq * C, kv_div, float(scale),
                                        _stream()), 'vsx_attention_f16')
    return out


def temporal_attention(q, k, v, B, fq, fk, hw, heads, scale):
    """q [B*fq*hw, C], k/v [B*fk*hw, C] in (b, f, site) row order (possibly column slices of one fused q|k|v
    buffer) -> [B*fq*hw, C]."""
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        _chk_view(t, n)
        if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
            raise _lib.VsxError(f'{n}: need a [rows, C] tensor with unit inner stride and aligned rows')
    if k.stride(0) != v.stride(0):
        raise _lib.VsxError('temporal_attention: k and v must share their row stride')
    C = q.shape[-1]
    out = torch.empty(q.shape[0], C, dtype=_F16, device=q.device)
    if FlopCounter.enabled:
        FlopCounter.attention += 4.0 * B * hw * fq * fk * C
    fn = _lib.load().vsx_temporal_attention_f16
    if fq <= 32 or fk > 128:
        check(fn(_p(q), _p(k), _p(v), _p(out), B, fq, fk, hw, heads, C // heads, q.stride(0), k.stride(0), C,
                 float(scale), _stream()), 'vsx_temporal_attention_f16')
        return out
    # more than 32 query frames (a long clip whose frames are all local: FrameShard exchange='sites'): the matrix-core
    # kernel takes up to 32 query frames against up to 128 key frames, so the queries go in blocks of 32 frames; a
    # block's rows are contiguous within one batch item, hence one launch per (batch item, block)
    for b in range(B):
        kb, vb = k[b * fk * hw:(b + 1) * fk * hw], v[b * fk * hw:(b + 1) * fk * hw]
        for f0 in range(0, fq, 32):
            n = min(32, fq - f0)
            r0 = (b * fq + f0) * hw
            check(fn(_p(q[r0:r0 + n * hw]), _p(kb), _p(vb), _p(out[r0:r0 + n * hw]), 1, n, fk, hw, heads, C // heads,
                     q.stride(0), k.stride(0), C, float(scale), _stream()), 'vsx_temporal_attention_f16')
    return out


# --------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------
def group_norm(x, gamma, beta, groups, eps, nimg, silu=False, x2=None, partial_hook=None, count_rows=None):
    """GroupNorm over channels-last x (viewed as [nimg, rows, C1]) (+x2 concatenated on C) -> [.., C1+C2].

    nimg = B reproduces the reference's 5-D GroupNorm (statistics over all frames), nimg = B*F the per-frame one.
    partial_hook(partial) may all-reduce the fp32 partial sums in frame-sharded mode (count_rows = global rows).
    """
    _chk(x, 'x'); _chk(x2, 'x2'); _chk(gamma, 'gamma'); _chk(beta, 'beta')
    C1 = x.shape[-1]
    C2 = x2.shape[-1] if x2 is not None else 0
    rows = x.numel() // C1 // nimg
    lib = _lib.load()
    nchunks = lib.vsx_groupnorm_chunks(rows, nimg)
    partial = torch.empty(nimg, nchunks, groups, 2, dtype=torch.float32, device=x.device)
    s = _stream()
    check(lib.vsx_groupnorm_stats(_p(x), _p(x2), nimg, rows, C1, C2, groups, _p(partial), s), 'vsx_groupnorm_stats')
    if partial_hook is not None:
        partial = partial_hook(partial)
        nchunks = partial.shape[1]
    y = torch.empty(*x.shape[:-1], C1 + C2, dtype=_F16, device=x.device)
    stats = torch.empty(nimg, groups, 2, dtype=torch.float32, device=x.device)
    check(lib.vsx_groupnorm_apply(_p(x), _p(x2), nimg, rows, C1, C2, groups, _p(partial), nchunks,
                                  rows if count_rows is None else count_rows, _p(gamma), _p(beta), float(eps),
                                  1 if silu else 0, _p(stats), _p(y), s), 'vsx_groupnorm_apply')
    return y


def layer_norm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0, frame_offset=0):
    if isinstance(x, DeferredLN):
        raise _lib.VsxError('layer_norm of a deferred LayerNorm: materialize() it first')
    _chk(x, 'x'); _chk(gamma, 'gamma'); _chk(beta, 'beta'); _chk(pe, 'pe')
    C = x.shape[-1]
    if pe is not None:
        # the kernel indexes pe[(row / rows_per_frame) % frames + frame_offset]: a clip longer than the table
        # (temporal_position_encoding_max_len) must fail like the reference's shape error, not read past the table
        if pe.dim() != 2 or pe.shape[1] != C or frames <= 0 or rows_per_frame <= 0 or frame_offset < 0 \
                or pe.shape[0] < frame_offset + frames:
            raise _lib.VsxError(f'layer_norm: positional-encoding table {tuple(pe.shape)} does not cover frames '
                                f'[{frame_offset}, {frame_offset + frames}) x {C} channels')
    y = torch.empty_like(x)
    check(_lib.load().vsx_layernorm(_p(x), x.numel() // C, C, _p(gamma), _p(beta), float(eps), _p(pe), rows_per_frame,
                                    frames, frame_offset, _p(y), _stream()), 'vsx_layernorm')
    return y


# --------------------------------------------------------------------------------------------
# element-wise glue
# --------------------------------------------------------------------------------------------
def silu(x):
    _chk(x, 'x')
    y = torch.empty_like(x)
    check(_lib.load().vsx_silu(_p(x), _p(y), x.numel(), _stream()), 'vsx_silu')
    return y


def quick_gelu(x):
    _chk(x, 'x')
    y = torch.empty_like(x)
    check(_lib.load().vsx_quick_gelu(_p(x), _p(y), x.numel(), _stream()), 'vsx_quick_gelu')
    return y


def axpy(a, b, s=1.0):
    _chk(a, 'a'); _chk(b, 'b')
    if a.shape != b.shape:
        raise _lib.VsxError(f'axpy: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')
    y = torch.empty_like(a)
    check(_lib.load().vsx_axpy(_p(a), _p(b), float(s), _p(y), a.numel(), _stream()), 'vsx_axpy')
    return y


def pack_latents(x, cpad=8):
    """[B,C,F,H,W] -> channels-last [B*F,H,W,cpad] (zero-padded channels)."""
    _chk(x, 'x')
    B, C, F, H, W = x.shape
    y = torch.empty(B * F, H, W, cpad, dtype=_F16, device=x.device)
    check(_lib.load().vsx_pack_latents(_p(x), _p(y), B, C, F, H * W, cpad, _stream()), 'vsx_pack_latents')
    return y


def unpack_latents(x, B, cout):
    """channels-last [B*F,H,W,Cs] -> [B,cout,F,H,W]."""
    _chk(x, 'x')
    BF, H, W, Cs = x.shape
    F = BF // B
    y = torch.empty(B, cout, F, H, W, dtype=_F16, device=x.device)
    check(_lib.load().vsx_unpack_latents(_p(x), _p(y), B, cout, F, H * W, Cs, _stream()), 'vsx_unpack_latents')
    return y


def cfg_ddim_step(x, eps_u, eps_c, guidance, alpha_t, alpha_next):
    _chk(x, 'x'); _chk(eps_u, 'eps_u'); _chk(eps_c, 'eps_c')
    out = torch.empty_like(x)
    check(_lib.load().vsx_cfg_ddim_step(_p(x), _p(eps_u), _p(eps_c), float(guidance), float(alpha_t),
                                        float(alpha_next), _p(out), x.numel(), _stream()), 'vsx_cfg_ddim_step')
    return out


def masked_blend(x, src, mask):
    """x, src [C, ...spatial]; mask [...spatial] -> src + mask*(x-src)."""
    _chk(x, 'x'); _chk(src, 'src'); _chk(mask, 'mask')
    C = x.shape[0]
    n_sp = x.numel() // C
    if mask.numel() != n_sp or src.shape != x.shape:
        raise _lib.VsxError('masked_blend: shape mismatch')
    out = torch.empty_like(x)
    check(_lib.load().vsx_masked_blend(_p(x), _p(src), _p(mask), _p(out), C, n_sp, _stream()), 'vsx_masked_blend')
    return out


def adapter_scatter(tracks, selected, feat, h, w, rate, out_scale=1.0):
    """tracks [F,P,2] fp32, selected [P] int32, feat [P,C] fp16 -> [F,h,w,C] fp16."""
    _chk(tracks, 'tracks', torch.float32); _chk(selected, 'selected', torch.int32); _chk(feat, 'feat')
    F, P = tracks.shape[:2]
    C = feat.shape[1]
    out = torch.zeros(F, h, w, C, dtype=_F16, device=feat.device)
    check(_lib.load().vsx_adapter_scatter(_p(tracks), _p(selected), _p(feat), _p(out), F, P, C, h, w, float(rate),
                                          float(out_scale), _stream()), 'vsx_adapter_scatter')
    return out


_options = {'gemm_pp': int(os.environ.get('VSX_GEMM_PP', '1')), 'tile_tune': int(os.environ.get('VSX_TUNE_TILE', '0'))}


def set_option(name, value):
    """Process-wide tuning / test switch of libvsx (see include/vsx.h: "gemm_pp", "pp_sched", "tile_tune").  The values are
    mirrored here for the host-side choices that depend on them (`_subpixel_eligible`)."""
    check(_lib.load().vsx_set_option(name.encode(), int(value)), 'vsx_set_option')
    _options[name] = int(value)


_prof_state = {'on': 0, 'max': 0, 'stride': 1, 'paused': False}


def prof_pause(paused):
    """Suspend / resume the hipEvent bracketing of GEMM launches (graph capture and graph-replayed calls)."""
    if paused == _prof_state['paused']:
        return
    _prof_state['paused'] = paused
    if _prof_state['on']:
        check(_lib.load().vsx_prof_pause(1 if paused else 0), 'vsx_prof_pause')


def prof_enable(on, max_samples=4096, stride=1):
    """Bracket every `stride`-th vsx_gemm_f16 launch with hipEvents (bench.py's roofline object)."""
    _prof_state.update(on=1 if on else 0, max=max_samples, stride=stride, paused=False)
    check(_lib.load().vsx_prof_enable((max(int(stride), 1) if on else 0), max_samples), 'vsx_prof_enable')


def prof_collect_roofline(peak_flops, peak_bytes_per_s):
    """-> dict(n, ms, flop, bytes, floor_ms, byte_bound_ms) over the sampled launches (include/vsx.h: vsx_prof_collect_roofline)"""
    n = ctypes.c_int64(0)
    v = [ctypes.c_double(0) for _ in range(5)]
    check(_lib.load().vsx_prof_collect_roofline(float(peak_flops), float(peak_bytes_per_s), ctypes.byref(n),
                                                *[ctypes.byref(x) for x in v]), 'vsx_prof_collect_roofline')
    return dict(n=n.value, ms=v[0].value, flop=v[1].value, bytes=v[2].value, floor_ms=v[3].value, byte_bound_ms=v[4].value)


def prof_collect():
    n = ctypes.c_int64(0)
    ms = ctypes.c_double(0)
    fl = ctypes.c_double(0)
    check(_lib.load().vsx_prof_collect(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), 'vsx_prof_collect')
    return n.value, ms.value, fl.value


# --------------------------------------------------------------------------------------------
# kernel functions of the gradient path (adapter training step): csrc/train.hip, declared in
# include/vsx.h since ABI 5 (first run on hardware in round 3)
# --------------------------------------------------------------------------------------------
def _train_fn(name):
    return getattr(_lib.load(), name)


def geglu_fwd(y2):
    """y2 [.., 2N] = (h | g) pre-activations -> h * gelu_erf(g) [.., N] (the GEMM's GEGLU epilogue as its own pass: the
    gradient needs the pre-activations)."""
    _chk(y2, 'y2')
    n = y2.shape[-1] // 2
    out = torch.empty(*y2.shape[:-1], n, dtype=_F16, device=y2.device)
    check(_train_fn('vsx_geglu_fwd')(_p(y2), _p(out), y2.numel() // (2 * n), n, _stream()), 'vsx_geglu_fwd')
    return out


def geglu_bwd(dout, y2):
    """-> d y2 [.., 2N]: dh = dout * gelu(g), dg = dout * h * gelu'(g)."""
    _chk(dout, 'dout'); _chk(y2, 'y2')
    n = y2.shape[-1] // 2
    dy2 = torch.empty_like(y2)
    check(_train_fn('vsx_geglu_bwd')(_p(dout), _p(y2), _p(dy2), y2.numel() // (2 * n), n, _stream()), 'vsx_geglu_bwd')
    return dy2


def silu_bwd(dy, x):
    _chk(dy, 'dy'); _chk(x, 'x')
    dx = torch.empty_like(x)
    check(_train_fn('vsx_silu_bwd')(_p(dy), _p(x), _p(dx), x.numel(), _stream()), 'vsx_silu_bwd')
    return dx


def group_norm_bwd(dy, x, gamma, beta, groups, eps, nimg, silu=False, x2=None):
    """Data gradient of `group_norm` (same arguments; dy [.., C1+C2]) -> (dx [.., C1], dx2 [.., C2] or None)."""
    _chk(dy, 'dy'); _chk(x, 'x'); _chk(x2, 'x2'); _chk(gamma, 'gamma'); _chk(beta, 'beta')
    C1 = x.shape[-1]
    C2 = x2.shape[-1] if x2 is not None else 0
    rows = x.numel() // C1 // nimg
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    ws = torch.empty(int(_train_fn('vsx_groupnorm_bwd_workspace')(nimg, rows, groups)), dtype=torch.float32, device=x.device)
    check(_train_fn('vsx_groupnorm_bwd')(_p(dy), _p(x), _p(x2), nimg, rows, C1, C2, groups, _p(gamma), _p(beta),
                                         float(eps), 1 if silu else 0, _p(ws), _p(dx), _p(dx2), _stream()),
          'vsx_groupnorm_bwd')
    return dx, dx2


def layer_norm_bwd(dy, x, gamma, eps=1e-5):
    """Data gradient of `layer_norm` (the positional encoding is added after the normalisation: it does not enter)."""
    _chk(dy, 'dy'); _chk(x, 'x'); _chk(gamma, 'gamma')
    C = x.shape[-1]
    dx = torch.empty_like(x)
    check(_train_fn('vsx_layernorm_bwd')(_p(dy), _p(x), _p(gamma), float(eps), _p(dx), x.numel() // C, C, _stream()),
          'vsx_layernorm_bwd')
    return dx


def softmax_bwd(probs, dprobs, scale):
    """dS = scale * P o (dP - rowsum(dP o P)), written over dP.  Both are views of row-padded buffers as
    `attention_scores` returns them ([nb, heads, nq, nk], row stride ld)."""
    nb, heads, nq, nk = probs.shape
    ld = probs.stride(2)
    if dprobs.shape != probs.shape or dprobs.stride() != probs.stride() or probs.stride(3) != 1:
        raise _lib.VsxError('softmax_bwd: probs and dprobs must share shape and (padded-row) strides')
    _chk_view(probs, 'probs'); _chk_view(dprobs, 'dprobs')
    check(_train_fn('vsx_softmax_bwd')(_p(probs), _p(dprobs), nb * heads * nq, nk, ld, float(scale), _stream()),
          'vsx_softmax_bwd')
    return dprobs


def attention_bwd_supported(dh):
    """Head dims the flash-attention backward kernels exist for (csrc/attention_bwd.hip)."""
    return bool(_train_fn('vsx_attention_bwd_supported')(int(dh)))


def attention_lse(q, k, vt, heads, scale, kv_div=1, nk=None):
    """`attention` that also returns lse [nb, heads, round_up(nq, 64)] fp32 (log2 of the softmax denominators in the
    scaled-score domain, zero beyond nq): what `attention_bwd` recomputes the probabilities from."""
    _chk(vt, 'vt')
    ldq, q_bs = _rows_view(q, 'q')
    ldk, k_bs = _rows_view(k, 'k')
    nb, nq, C = q.shape
    nk = k.shape[1] if nk is None else nk
    dh = C // heads
    out = torch.empty(nb, nq, C, dtype=_F16, device=q.device)
    lds = round_up(nq, 64)
    lse = torch.zeros(nb, heads, lds, dtype=torch.float32, device=q.device)
    if FlopCounter.enabled:
        FlopCounter.attention += 4.0 * nb * heads * nq * nk * dh
    check(_train_fn('vsx_attention_lse_f16')(_p(q), _p(k), _p(vt), _p(out), _p(lse), lds, nb, heads, nq, nk, dh, ldq, ldk,
                                             vt.shape[2], C, q_bs, k_bs, C * vt.shape[2], nq * C, kv_div, float(scale),
                                             _stream()), 'vsx_attention_lse_f16')
    return out, lse


def attention_bwd(q, k, v, out, dout, lse, heads, scale, kv_div=1, need_kv=True):
    """Data gradients of the fused attention without the [heads, nq, nk] probabilities (csrc/attention_bwd.hip).
    q, out, dout [nb, nq, C]; k, v [nb / kv_div, nk, C], all contiguous; lse from `attention_lse`.
    -> (dq, dk, dv) (dk = dv = None unless need_kv; kv_div > 1: dq only)."""
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v'), (out, 'out'), (dout, 'dout')):
        _chk(t, n)
    _chk(lse, 'lse', torch.float32)
    nb, nq, C = q.shape
    nk = k.shape[1]
    dh = C // heads
    lds = lse.shape[-1]

    def transposed(t):          # [n_img, n, C] -> [n_img, C, round_up(n, 8)], zero padded
        n = t.shape[1]
        ld = round_up(n, 8)
        tt = torch.zeros(t.shape[0], C, ld, dtype=_F16, device=t.device) if ld != n else \
            torch.empty(t.shape[0], C, ld, dtype=_F16, device=t.device)
        tt[:, :, :n] = t.transpose(1, 2)
        return tt
    kt = transposed(k)
    qt = dot = dk = dv = None
    if need_kv:
        if kv_div != 1:
            raise _lib.VsxError('attention_bwd: key / value gradients of a shared (text) context')
        qt, dot = transposed(q), transposed(dout)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
    dq = torch.empty_like(q)
    delta = torch.empty(nb, heads, lds, dtype=torch.float32, device=q.device)
    check(_train_fn('vsx_attention_bwd_f16')(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(qt), _p(kt), _p(dot), _p(lse),
                                             _p(delta), _p(dq), _p(dk), _p(dv), nb, heads, nq, nk, dh,
                                             qt.shape[2] if qt is not None else 0, kt.shape[2], lds, kv_div, float(scale),
                                             _stream()), 'vsx_attention_bwd_f16')
    return dq, dk, dv


def sum_pool2x2(x):
    """[n, 2h, 2w, c] -> [n, h, w, c]: the data gradient of the nearest-2x upsampling folded into a conv's loader."""
    _chk(x, 'x')
    n, h2, w2, c = x.shape
    y = torch.empty(n, h2 // 2, w2 // 2, c, dtype=_F16, device=x.device)
    check(_train_fn('vsx_sum_pool2x2')(_p(x), _p(y), n, h2 // 2, w2 // 2, c, _stream()), 'vsx_sum_pool2x2')
    return y


def adapter_gather(tracks, selected, dmap, rate, out_scale=1.0):
    """Gradient of `adapter_scatter` with respect to feat: dfeat[p, :] = out_scale * sum over frames and the 4 bilinear
    corners of weight * dmap[f, y, x, :] (same fp16 sub-pixel positions and weights as the forward)."""
    _chk(tracks, 'tracks', torch.float32); _chk(selected, 'selected', torch.int32); _chk(dmap, 'dmap')
    F, P = tracks.shape[:2]
    _, h, w, C = dmap.shape
    dfeat = torch.zeros(P, C, dtype=_F16, device=dmap.device)
    check(_train_fn('vsx_adapter_gather')(_p(tracks), _p(selected), _p(dmap), _p(dfeat), F, P, C, h, w, float(rate),
                                          float(out_scale), _stream()), 'vsx_adapter_gather')
    return dfeat


# --------------------------------------------------------------------------------------------
# kernel table and gradient dispatch
# --------------------------------------------------------------------------------------------
# Every op above is a "kernel function" (it launches HIP kernels and nothing else).  The names the rest of the package
# calls are thin wrappers that (1) look the kernel function up in `_raw` at call time — one table a test harness can
# swap as a whole — and (2) when autograd is recording and an ACTIVATION argument requires grad (only the adapter
# training step, trainer_videoswap.py:33-97: the UNet's weights are frozen), route the call through the matching
# torch.autograd.Function of videoswap_amd/autograd.py, whose backward is again built from kernel functions.
_raw = {}

# op name -> positions / keywords of the arguments whose requires_grad switches the gradient path on: the activation
# arguments, and for `linear` also weight and bias (the adapter's MLPs are the trainable parameters of the training step:
# their first Linear gets an input WITHOUT grad — the point embedding — and must still record, adapter_model.py:70-107)
_ACTIVATIONS = {
    'linear': ((0, 1, 2), ('weight', 'bias', 'residual')), 'linear_vt': ((0,), ()), 'conv2d': ((0,), ('x2', 'residual')),
    'attention': ((0, 1, 2), ()), 'temporal_attention': ((0, 1, 2), ()), 'attention_scores': ((0, 1), ()),
    'attention_pv': ((0, 1), ()), 'head_scores': ((0, 1), ()), 'group_norm': ((0,), ('x2',)), 'layer_norm': ((0,), ()),
    'silu': ((0,), ()), 'quick_gelu': ((0,), ()), 'axpy': ((0, 1), ()), 'pack_latents': ((0,), ()),
    'unpack_latents': ((0,), ()), 'cfg_ddim_step': ((0, 1, 2), ()), 'masked_blend': ((0, 1), ()),
    'adapter_scatter': ((2,), ()),
}
_PLAIN = ('gemm', 'set_option', 'prof_pause', 'prof_enable', 'prof_collect', 'prof_collect_roofline', 'geglu_fwd', 'geglu_bwd', 'silu_bwd',
          'group_norm_bwd', 'layer_norm_bwd', 'softmax_bwd', 'sum_pool2x2', 'adapter_gather', 'attention_lse',
          'attention_bwd', 'attention_bwd_supported')


def _publish(name):
    _raw[name] = globals()[name]
    pos, kws = _ACTIVATIONS.get(name, ((), ()))

    def op(*args, **kwargs):
        if pos and torch.is_grad_enabled():
            for i in pos:
                if i < len(args) and torch.is_tensor(args[i]) and args[i].requires_grad:
                    break
            else:
                for k in kws:
                    t = kwargs.get(k)
                    if t is not None and t.requires_grad:
                        break
                else:
                    return _raw[name](*args, **kwargs)
            from . import autograd
            # the gradient path works on plain tensors: a LayerNorm deferred into this consumer (its own input had no
            # gradient, e.g. unfrozen weights under grad mode) is applied by its standalone kernel first
            if args and isinstance(args[0], DeferredLN):
                args = (args[0].materialize(),) + tuple(args[1:])
            return autograd.dispatch(name, *args, **kwargs)
        return _raw[name](*args, **kwargs)

    op.__name__ = name
    op.__doc__ = _raw[name].__doc__
    globals()[name] = op


for _name in tuple(_ACTIVATIONS) + _PLAIN:
    _publish(_name)
del _name
