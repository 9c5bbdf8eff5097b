"""Frame-sharded long-clip mode (BASELINE.json configs[3]) on ONE GPU: two processes share cuda:0, each owns half the
frames, the temporal K/V all-gather and the GroupNorm partial-sum exchange go through gloo (staged via the host; on a
multi-GPU node the same hooks run on RCCL).  The gathered result must match the single-device oracle on the full clip."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, HW = 8, 16


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    import traceback
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from util import oracle_unet, product_unet_from
        from oracle import unet3d
        from videoswap_amd.distributed import FrameShard
        cfg = unet3d.tiny_config()
        ora = oracle_unet(cfg)
        prod = product_unet_from(ora, cfg)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 4, T, HW, HW, generator=g)
        txt = torch.randn(2, 77, 64, generator=g)
        shard = FrameShard(T)
        shard.install(prod)
        with torch.no_grad():
            local = prod(shard.local_slice(x).half().cuda(), 301, txt.half().cuda()).sample
            full = shard.gather_frames(local).float().cpu()
        FrameShard.uninstall(prod)
        out = None
        if rank == 0:
            with torch.no_grad():
                ref = ora(x, torch.tensor(301), txt).sample
                half = prod(x[:, :, :T // 2].half().cuda(), 301, txt.half().cuda()).sample.float().cpu()
            err = float((full - ref).norm() / ref.norm())
            # without the exchange the first half differs visibly (cross-frame coupling is real)
            uncoupled = float((half - ref[:, :, :T // 2]).norm() / ref[:, :, :T // 2].norm())
            out = (err, uncoupled)
        dist.barrier()
        dist.destroy_process_group()
        q.put(('ok', rank, out))
    except Exception:  # pragma: no cover
        q.put(('error', rank, traceback.format_exc()))


def test_frame_sharded_unet_matches_full_clip_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
    errs = [r for r in results if r[0] != 'ok']
    assert not errs, errs[0][2]
    err, uncoupled = [r[2] for r in results if r[1] == 0][0]
    print(f'frame-sharded (2 x {T // 2} frames) vs full-clip oracle: rel-L2 {err:.3e}; unsharded half-clip {uncoupled:.3e}')
    assert err < 4e-3
    assert uncoupled > 3 * err
