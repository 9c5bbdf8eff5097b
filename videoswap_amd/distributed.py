"""Multi-GPU modes of the denoising path: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests and by single-GPU emulation).

1. Clip-parallel (BASELINE.json configs[4]): independent clips are sharded round-robin over the ranks; there is NO
   collective on the data path (`shard_clips`, `max_over_ranks` only serve the benchmark bookkeeping).

2. Frame-sharded long-clip mode (configs[3]): rank r owns frames [r*T/P, (r+1)*T/P) of the [B, F, H, W, C] activations;
   weights are replicated.  Everything in the UNet is per-frame local except two couplings, which `FrameShard` wires in:
     * temporal attention attends across ALL frames: the fused K|V projection of the local frames is all-gathered over
       the frame axis into one [B, F_total, hw, 2C] buffer (`kv_gather_start/finish`: one collective per batch item,
       in flight while the q projection runs), consumed by `vsx_temporal_attention_f16` with fq local / fk global
       frames (its long-clip MFMA kernel); the positional encoding uses the GLOBAL frame index (`frame_offset`);
     * the reference's ResnetBlock3D / conv_norm_out GroupNorm pools statistics over frames (resnet.py:166,177;
       unet.py:474): the fp32 partial sums produced by `vsx_groupnorm_stats` are all-gathered (`gn_hook`) and reduced
       in rank order by `vsx_groupnorm_apply`, so every rank computes bit-identical statistics.
   `FrameShard(..., exchange='sites')` replaces the 40 K|V all-gathers by TWO all-to-alls per motion module: after the
   module's per-frame GroupNorm and `proj_in` the activation is re-sharded from frames to SITES (rank r receives the
   rows of sites [r*hw/P, (r+1)*hw/P) of EVERY frame), the whole temporal transformer (LayerNorm + PE, both temporal
   attentions, feed-forward: all of it local to a site) runs unchanged on [B, F_total, hw/P, C], and the result is
   re-sharded back in front of `proj_out` (+ residual).  A rank then moves 2 * (P-1)/P of its OWN activation per motion
   module instead of receiving (P-1) * 2C columns of everybody's K|V twice: 16 x fewer bytes at P = 8.
   The reference cannot run T > 24 at all (PositionalEncoding max_len 24, motion_module.py:237-255): the long-clip
   model is the same architecture with `temporal_position_encoding_max_len` extended (closed-form sinusoid).
"""
import torch
import torch.distributed as dist


def shard_clips(n_clips, rank, world):
    """Indices of the clips rank `rank` processes (round-robin: clips are independent units)."""
    return list(range(rank, n_clips, world))


def max_over_ranks(seconds, device=None):
    """Whole-job time = slowest rank (bench.py contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or _coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _coll_device():
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def _all_gather_into(out, t, group=None):
    """`out` [world, *t.shape] <- every rank's `t`, in rank order: ONE collective into a preallocated tensor (no list
    of parts, no torch.cat).  With gloo (CPU tests, single-GPU emulation) GPU tensors are staged through the host."""
    assert out.is_contiguous() and out.numel() == dist.get_world_size(group) * t.numel()
    if dist.get_backend(group) == 'nccl' or not t.is_cuda:
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=group)
        return out
    host_out = torch.empty(out.numel(), dtype=out.dtype)
    dist.all_gather_into_tensor(host_out, t.detach().cpu().contiguous().view(-1), group=group)
    out.view(-1).copy_(host_out)
    return out


def _all_to_all(out, t, group=None):
    """`t` [world, n] -> `out` [world, n]: block p of `t` goes to rank p, block s of `out` comes from rank s.  ONE
    collective; with gloo, GPU tensors are staged through the host (CPU tests, single-GPU emulation)."""
    assert out.is_contiguous() and t.is_contiguous() and out.numel() == t.numel()
    if dist.get_backend(group) == 'nccl' or not t.is_cuda:
        dist.all_to_all_single(out.view(-1), t.view(-1), group=group)
        return out
    host_out = torch.empty(out.numel(), dtype=out.dtype)
    dist.all_to_all_single(host_out, t.detach().cpu().view(-1), group=group)
    out.view(-1).copy_(host_out)
    return out


class RcclComm:
    """The library's own RCCL communicator (include/vsx.h: vsx_comm_*): one per process.  The 128-byte unique id is
    created by rank 0 and broadcast through torch.distributed's store-backed object collective (out-of-band: any
    rendezvous would do)."""

    def __init__(self, group=None):
        import ctypes
        from . import _lib
        self.lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(self.lib.vsx_comm_unique_id(buf), 'vsx_comm_unique_id')
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = ctypes.create_string_buffer(box[0], 128)
        _lib.check(self.lib.vsx_comm_init(self.rank, self.world, uid), 'vsx_comm_init')
        self.stream = torch.cuda.Stream()        # collectives run beside the compute stream

    def close(self):
        from . import _lib
        _lib.check(self.lib.vsx_comm_destroy(), 'vsx_comm_destroy')


class FrameShard:
    """Frame-axis sharding of one clip over the ranks of `group` (see module docstring).

    `backend='rccl'` routes the two exchanges through the library's C-ABI collectives (vsx_allgather_kv /
    vsx_allgather_f32) on a side stream: the K|V all-gather of a temporal attention overlaps with its q projection.
    `backend='torch'` uses torch.distributed (`all_gather_into_tensor`; gloo in the CPU tests and the single-GPU
    emulation).  Either way each exchange is ONE collective per batch item into a preallocated buffer laid out as the
    attention kernel reads it ([B, F_total, hw, 2C]: K = columns [0, C), V = [C, 2C)).

    `exchange='kv'` gathers the temporal K|V; `exchange='sites'` re-shards the activation frames <-> sites around every
    motion module instead (module docstring): `to_sites` / `to_frames`, one all-to-all each (torch.distributed, or the
    library's vsx_alltoall_f16 with backend 'rccl').  The default `'auto'` decides per motion module: the site re-shard
    (4x fewer bytes at 2 ranks, 16x at 8; both forms measured on an MI355X in round 3: same rel-L2 against the full-clip
    oracle, 190 vs 760 MB received per rank and forward at 2 x 32 frames) wherever the level's site count splits over
    the ranks, the K|V all-gather where it does not (e.g. the 7 x 12 level of a 448 x 768 clip on 8 ranks)."""

    def __init__(self, total_frames, group=None, backend=None, exchange='auto'):
        if exchange not in ('auto', 'kv', 'sites'):
            raise ValueError("exchange must be 'auto', 'kv' (all-gather of the temporal K|V) or 'sites' (frame <-> site "
                             "all-to-all around every motion module)")
        self.exchange = exchange
        self.kv_active = True                    # switched off while a motion module runs on the site layout
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if total_frames % self.world:
            raise ValueError(f'{total_frames} frames do not split evenly over {self.world} ranks')
        self.total_frames = total_frames
        self.local_frames = total_frames // self.world
        self.frame_offset = self.rank * self.local_frames
        if backend is None:
            backend = 'rccl' if dist.get_backend(group) == 'nccl' else 'torch'
        self.backend = backend
        self.comm = RcclComm(group) if backend == 'rccl' else None
        self.bytes_gathered = 0                  # bytes received from the other ranks (K|V rows or re-sharded
                                                 # activations) since construction

    # ---- GroupNorm partial sums ----------------------------------------------------------------------------------
    def gn_hook(self, partial):
        """[B, nchunks, groups, 2] fp32 partial sums of the local frames -> [B, world*nchunks, groups, 2] of all
        frames in rank order (every rank then reduces them in the same fixed order: bit-identical statistics)."""
        b, nchunks, groups, two = partial.shape
        gathered = torch.empty(self.world, b, nchunks, groups, two, dtype=partial.dtype, device=partial.device)
        if self.comm is not None:
            from . import _lib, ops
            part = partial.contiguous()
            _lib.check(self.comm.lib.vsx_allgather_f32(ops._p(part), ops._p(gathered), part.numel(), ops._stream()),
                       'vsx_allgather_f32')
        else:
            _all_gather_into(gathered, partial, self.group)
        return gathered.permute(1, 0, 2, 3, 4).reshape(b, self.world * nchunks, groups, two).contiguous()

    # ---- temporal K|V ----------------------------------------------------------------------------------------------
    def kv_gather_start(self, kv, b, frames, hw):
        """kv [b*frames*hw, 2C] (this rank's frames, K | V columns) -> handle; the collective is in flight when this
        returns (rccl backend) so that the caller can launch the q projection behind it."""
        c2 = kv.shape[-1]
        per_batch = frames * hw * c2
        out = torch.empty(b, self.world, per_batch, dtype=kv.dtype, device=kv.device)
        self.bytes_gathered += (self.world - 1) * b * per_batch * kv.element_size()
        if self.comm is not None:
            from . import _lib, ops
            cur = torch.cuda.current_stream()
            side = self.comm.stream
            side.wait_stream(cur)                                # kv (and `out`'s allocation) are ready
            kv.record_stream(side)
            out.record_stream(side)
            import ctypes
            with torch.cuda.stream(side):
                _lib.check(self.comm.lib.vsx_allgather_kv(ops._p(kv), ops._p(out), b, per_batch,
                                                          ctypes.c_void_p(side.cuda_stream)), 'vsx_allgather_kv')
                done = torch.cuda.Event()
                done.record(side)
            return out, done, c2
        if b == 1:
            _all_gather_into(out.view(self.world, per_batch), kv.reshape(per_batch), self.group)
        else:                                    # the frame axis is not outermost: one collective per batch item
            for i in range(b):
                part = torch.empty(self.world, per_batch, dtype=kv.dtype, device=kv.device)
                _all_gather_into(part, kv.view(b, per_batch)[i], self.group)
                out[i] = part
        return out, None, c2

    def kv_gather_finish(self, handle):
        """-> (kv_all [b*F_total*hw, 2C], F_total)"""
        out, done, c2 = handle
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        return out.view(-1, c2), self.total_frames

    # ---- frames <-> sites (exchange = 'sites') ---------------------------------------------------------------------
    def use_sites(self, hw):
        """Does the motion module at a level with `hw` sites run on the site layout?"""
        if self.exchange == 'kv':
            return False
        if self.exchange == 'sites':
            self.sites_per_rank(hw)              # raises when the level does not split
            return True
        return hw % self.world == 0

    def sites_per_rank(self, hw):
        if hw % self.world:
            raise ValueError(f"exchange='sites' needs every level's site count to split over the ranks: {hw} sites, "
                             f"{self.world} ranks (use exchange='kv' for this latent size)")
        return hw // self.world

    def _alltoall_strided(self, src, dst, b, block, send_strides, recv_strides):
        """RCCL path of both re-shards through the C ABI (vsx_alltoall_f16: grouped send / recv straight from / into the
        strided layouts, no pack / unpack passes)."""
        import ctypes
        from . import _lib, ops
        arr = ctypes.c_int64 * 3
        _lib.check(self.comm.lib.vsx_alltoall_f16(ops._p(src), ops._p(dst), b, self.local_frames, block,
                                                  arr(*send_strides), arr(*recv_strides), ops._stream()),
                   'vsx_alltoall_f16')

    @staticmethod
    def reshard_strides(world, local_frames, block):
        """(send, recv) stride triples (peer, batch item, frame; elements) of the frames -> sites all-to-all between the
        layouts [b, f, P, hw/P, C] and [b, P, f, hw/P, C] (`block` = hw/P * C); sites -> frames swaps the two."""
        f, p = local_frames, world
        return (block, f * p * block, p * block), (f * block, p * f * block, block)

    def _has_c_alltoall(self):
        return self.comm is not None and getattr(self.comm.lib, 'vsx_alltoall_f16', None) is not None

    def to_sites(self, y, b, hw):
        """y [b*f_local*hw, C] (this rank's frames, all sites) -> [b*F_total*hw_local, C] (all frames, this rank's
        sites), rows in (b, frame, site) order with the frames in rank order = global order.  One all-to-all."""
        c = y.shape[-1]
        f, p = self.local_frames, self.world
        hl = self.sites_per_rank(hw)
        blk = hl * c
        self.bytes_gathered += (p - 1) * b * f * blk * y.element_size()
        if self._has_c_alltoall():
            out = torch.empty(b * p * f * hl, c, dtype=y.dtype, device=y.device)          # [b, p(source), f, hl, c]
            send_st, recv_st = self.reshard_strides(p, f, blk)
            self._alltoall_strided(y, out, b, blk, send_st, recv_st)
            return out
        # destination-major send blocks [p][b, f, hl, c]: one strided copy on each side of the collective
        send = y.view(b, f, p, hl, c).permute(2, 0, 1, 3, 4).contiguous()
        recv = torch.empty_like(send)                                    # [source rank s][b, f(s), hl, c]
        _all_to_all(recv.view(p, -1), send.view(p, -1), self.group)
        # frames of source s are global frames [s*f, (s+1)*f): (s, b, f) -> (b, s, f)
        return recv.permute(1, 0, 2, 3, 4).reshape(b * p * f * hl, c)

    def to_frames(self, ys, b, hw):
        """inverse of `to_sites`: ys [b*F_total*hw_local, C] -> [b*f_local*hw, C]."""
        c = ys.shape[-1]
        f, p = self.local_frames, self.world
        hl = self.sites_per_rank(hw)
        blk = hl * c
        self.bytes_gathered += (p - 1) * b * f * blk * ys.element_size()
        if self._has_c_alltoall():
            out = torch.empty(b * f * hw, c, dtype=ys.dtype, device=ys.device)            # [b, f, p(source), hl, c]
            recv_st, send_st = self.reshard_strides(p, f, blk)
            self._alltoall_strided(ys, out, b, blk, send_st, recv_st)
            return out
        send = ys.view(b, p, f, hl, c).permute(1, 0, 2, 3, 4).contiguous()   # block d = frames owned by rank d
        recv = torch.empty_like(send)                                        # [source s][b, f, hl(s), c]
        _all_to_all(recv.view(p, -1), send.view(p, -1), self.group)
        # sites of source s are [s*hl, (s+1)*hl): (s, b, f, hl) -> (b, f, s, hl)
        return recv.permute(1, 2, 0, 3, 4).reshape(b * f * hw, c)

    # ---- wiring ------------------------------------------------------------------------------------------------
    def install(self, unet):
        """Attach the hooks to a videoswap_amd AnimateDiffUNet3DModel (idempotent)."""
        from .attention import VanillaAttentionProcessor
        unet._frame_shard = self
        for m in unet.modules():
            proc = getattr(m, 'processor', None)
            if isinstance(proc, VanillaAttentionProcessor):
                if self.exchange != 'sites':     # 'sites': the temporal transformer sees every frame, no hook needed
                    proc.kv_gather = self        # ('auto': consulted only while `kv_active`)
                if proc.pos_encoder is not None and proc.pos_encoder.pe.shape[1] < self.total_frames:
                    raise ValueError('temporal_position_encoding_max_len is smaller than the clip: build the UNet with '
                                     f'max_len >= {self.total_frames}')
        return unet

    @staticmethod
    def uninstall(unet):
        from .attention import VanillaAttentionProcessor
        unet._frame_shard = None
        for m in unet.modules():
            proc = getattr(m, 'processor', None)
            if isinstance(proc, VanillaAttentionProcessor):
                proc.kv_gather = None

    def local_slice(self, latents):
        """[B, C, F, H, W] -> this rank's frames"""
        return latents[:, :, self.frame_offset:self.frame_offset + self.local_frames].contiguous()

    def gather_frames(self, latents_local):
        """this rank's [B, C, f, H, W] -> full clip on every rank"""
        x = latents_local.contiguous()
        parts = torch.empty(self.world, *x.shape, dtype=x.dtype, device=x.device)
        _all_gather_into(parts, x, self.group)
        b, c, f, h, w = x.shape
        return parts.permute(1, 2, 0, 3, 4, 5).reshape(b, c, self.world * f, h, w).contiguous()
