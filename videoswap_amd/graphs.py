"""HIP-graph replay of the UNet forward.

One graph per input signature (B, F, H, W, text shape, adapter-residual shapes) and (weights epoch, processor epoch).
A graph owns static input buffers (latents, the SiLU'd time-embedding row, the text embedding, the adapter maps) and
its output buffer; a call copies the inputs in, replays, and returns a copy of the output.  The captured body
contains kernel launches only; the step-invariant host caches (text K/V, time-embedding projections) are bypassed
inside a graph — the static buffers are new tensor objects, so the projections are captured as kernels of the graph —
which makes a graph independent of cache lifetimes (bench.py clears the caches per clip).

torch.cuda.CUDAGraph is hipGraph on ROCm; capture, memory pool and replay are PyTorch plumbing, the nodes are the
libvsx kernels launched through the C ABI on the capturing stream.
"""
import torch

from . import ops


class _Entry:
    __slots__ = ('graph', 'sample', 'semb', 'text', 'residuals', 'out', 'flop_gemm', 'flop_attention', 'flop_gemm_saved', 'keep')


class GraphCache:
    def __init__(self, eager_every=0, max_entries=8):
        self.eager_every = int(eager_every)
        self.max_entries = max_entries
        self.entries = {}
        self.calls = 0
        self.replays = 0
        self.captures = 0
        self.on_eager = None          # optional callable(bool): bench.py switches its launch profiler with it

    def _key(self, unet, sample, text, residuals, half):
        from .attention import Attention
        res = None if residuals is None else tuple(tuple(r.shape) for r in residuals)
        # `half` (the shared CFG prefix, decided by AnimateDiffUNet3DModel.forward on the caller's tensor) is part of the
        # signature: a graph captured with the prefix shared must never be replayed for a batch whose halves differ
        return (tuple(sample.shape), tuple(text.shape), res, int(half), sample.dtype, str(sample.device), unet._weights_epoch,
                Attention.processor_epoch)

    def run(self, unet, sample, silu_emb, text, residuals, half=0):
        self.calls += 1
        if self.eager_every and self.calls % self.eager_every == 0:
            if self.on_eager:
                self.on_eager(True)
            try:
                return unet._forward_body(sample, silu_emb, text, residuals, half)
            finally:
                if self.on_eager:
                    self.on_eager(False)
        key = self._key(unet, sample, text, residuals, half)
        e = self.entries.get(key)
        if e is None:
            stale = [k for k in self.entries if k[-2:] != key[-2:]]      # other weights / processors: dead graphs
            for k in stale:
                del self.entries[k]
            if len(self.entries) >= self.max_entries:
                self.entries.clear()
            e = self.entries[key] = self._capture(unet, sample, silu_emb, text, residuals, half)
        e.sample.copy_(sample)
        e.semb.copy_(silu_emb)
        if e.text.data_ptr() != text.data_ptr():
            e.text.copy_(text)
        if residuals is not None:
            for dst, src in zip(e.residuals, residuals):
                dst.copy_(src)
        e.graph.replay()
        self.replays += 1
        if ops.FlopCounter.enabled:      # the replayed launches do not pass through the Python wrappers
            ops.FlopCounter.gemm += e.flop_gemm
            ops.FlopCounter.attention += e.flop_attention
            ops.FlopCounter.gemm_saved += e.flop_gemm_saved
        return e.out.clone()

    def _capture(self, unet, sample, silu_emb, text, residuals, half=0):
        e = _Entry()
        e.keep = None
        e.sample, e.semb, e.text = sample.clone(), silu_emb.clone(), text.clone()
        e.residuals = None if residuals is None else [r.clone() for r in residuals]
        # one eager pass on a side stream first: every kernel's one-time host setup (LDS size attributes, device
        # queries, split-K workspace growth) must have happened before the capture
        from .layers import StepInvariantCache
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # no step-invariant host cache may answer for the static buffers: a projection cached on `e.semb` / `e.text` by the
        # warm-up pass would be missing from the captured graph, and every replay would reuse the first call's time-
        # embedding and text projections (found when the graph tests joined the default GPU suite in round 4)
        StepInvariantCache.bypass = True
        try:
            with torch.cuda.stream(side):
                unet._forward_body(e.sample, e.semb, e.text, e.residuals, half)
        except BaseException:
            StepInvariantCache.bypass = False
            raise
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.prof_pause(True)          # hipEvent pairs cannot be recorded inside a capture
        fc = ops.FlopCounter
        saved = (fc.enabled, fc.gemm, fc.attention, fc.gemm_saved)
        fc.enabled, fc.gemm, fc.attention, fc.gemm_saved = True, 0.0, 0.0, 0.0
        try:
            e.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.graph):
                e.out = unet._forward_body(e.sample, e.semb, e.text, e.residuals, half)
            # the graph's kernel arguments point at the folded LayerNorm operands of ops._fold_cache: hold them
            e.keep = ops.fold_cache_tensors()
        finally:
            StepInvariantCache.bypass = False
            e.flop_gemm, e.flop_attention, e.flop_gemm_saved = fc.gemm, fc.attention, fc.gemm_saved
            fc.enabled, fc.gemm, fc.attention, fc.gemm_saved = saved
            ops.prof_pause(False)
        self.captures += 1
        return e
