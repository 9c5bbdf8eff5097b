"""Multi-GPU modes of the denoising path: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests and by single-GPU emulation).

1. Clip-parallel (BASELINE.json configs[4]): independent clips are sharded round-robin over the ranks; there is NO
   collective on the data path (`shard_clips`, `max_over_ranks` only serve the benchmark bookkeeping).

2. Frame-sharded long-clip mode (configs[3]): rank r owns frames [r*T/P, (r+1)*T/P) of the [B, F, H, W, C] activations;
   weights are replicated.  Everything in the UNet is per-frame local except two couplings, which `FrameShard` wires in:
     * temporal attention attends across ALL frames: K and V of the local frames are all-gathered over the frame axis
       right after their projections (`kv_gather`, consumed by `vsx_temporal_attention_f16` with fq local / fk global
       frames); the positional encoding uses the GLOBAL frame index (`frame_offset`);
     * the reference's ResnetBlock3D / conv_norm_out GroupNorm pools statistics over frames (resnet.py:166,177;
       unet.py:474): the fp32 partial sums produced by `vsx_groupnorm_stats` are all-gathered (`gn_hook`) and reduced
       in rank order by `vsx_groupnorm_apply`, so every rank computes bit-identical statistics.
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


def _all_gather(t, group=None):
    """all-gather of equal-shape tensors -> list in rank order.  With gloo (CPU tests, single-GPU emulation) GPU
    tensors are staged through the host."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == 'nccl' or not t.is_cuda:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous(), group=group)
        return out
    host = t.detach().cpu().contiguous()
    out = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(out, host, group=group)
    return [o.to(t.device) for o in out]


class FrameShard:
    """Frame-axis sharding of one clip over the ranks of `group` (see module docstring)."""

    def __init__(self, total_frames, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if total_frames % self.world:
            raise ValueError(f'{total_frames} frames do not split evenly over {self.world} ranks')
        self.total_frames = total_frames
        self.local_frames = total_frames // self.world
        self.frame_offset = self.rank * self.local_frames

    # ---- hooks -------------------------------------------------------------------------------------------------
    def gn_hook(self, partial):
        """[B, nchunks, groups, 2] fp32 partial sums of the local frames -> [B, world*nchunks, groups, 2] of all
        frames, concatenated in rank order (deterministic reduction order on every rank)."""
        return torch.cat(_all_gather(partial, self.group), dim=1).contiguous()

    def kv_gather(self, k, v, b, frames, hw):
        """k, v [b*frames, hw, C] of the local frames -> [b*total_frames, hw, C] (frame-major per batch) and the
        global frame count."""
        c = k.shape[-1]

        def gather(t):
            parts = _all_gather(t.view(b, frames, hw, c), self.group)          # world x [b, f_local, hw, c]
            return torch.cat(parts, dim=1).reshape(b * self.total_frames, hw, c).contiguous()
        return gather(k), gather(v), self.total_frames

    # ---- wiring ------------------------------------------------------------------------------------------------
    def install(self, unet):
        """Attach the hooks to a videoswap_amd AnimateDiffUNet3DModel (idempotent)."""
        from .attention import VanillaAttentionProcessor
        unet._frame_shard = self
        for m in unet.modules():
            proc = getattr(m, 'processor', None)
            if isinstance(proc, VanillaAttentionProcessor):
                proc.kv_gather = self.kv_gather
                proc.frame_offset = self.frame_offset
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
                proc.frame_offset = 0

    def local_slice(self, latents):
        """[B, C, F, H, W] -> this rank's frames"""
        return latents[:, :, self.frame_offset:self.frame_offset + self.local_frames].contiguous()

    def gather_frames(self, latents_local):
        """this rank's [B, C, f, H, W] -> full clip on every rank"""
        return torch.cat(_all_gather(latents_local, self.group), dim=2)
