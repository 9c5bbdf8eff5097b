"""Host side of the product on CPU (always CPU, also on a GPU box): the HIP-backed UNet with `videoswap_amd.ops` swapped
for the plain-PyTorch restatement of the ops CONTRACT in tests/host_emulation.py (test infrastructure — see its header;
the product itself has no CPU path, which the first test asserts).  Checked here: the frame-sharded long-clip mode
inside the real UNet, both `exchange` forms, two ranks over gloo, against the full-clip oracle and the unsharded
product.  (The model-level tests marked `device` — test_unet_gpu, test_pipeline_gpu, test_processors_gpu,
test_swap_flow_gpu, test_config, test_vae, test_clip — use the same emulation when there is no GPU.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from util import oracle_unet, product_unet_from, rel_l2


def test_product_ops_refuse_cpu_tensors():
    from videoswap_amd import _lib, ops
    with pytest.raises(_lib.VsxError):
        ops.silu(torch.zeros(8, dtype=torch.float16))
    with pytest.raises(_lib.VsxError):
        ops.linear(torch.zeros(4, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _shard_worker(rank, world, port, q, exchange):
    import sys
    import traceback
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        torch.set_num_threads(2)
        import host_emulation as he
        from oracle import unet3d
        from videoswap_amd.distributed import FrameShard
        # 'auto' on a 24x24 latent: levels of 576 / 144 / 36 / 9 sites — the 3x3 level does not split over 2 ranks and
        # falls back to the K|V all-gather, the others take the site re-shard
        T, HW = 8, (24 if exchange == 'auto' else 16)
        cfg = unet3d.tiny_config()
        ora = oracle_unet(cfg)
        prod = product_unet_from(ora, cfg, device='cpu')
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 4, T, HW, HW, generator=g)
        txt = torch.randn(2, 77, 64, generator=g)
        shard = FrameShard(T, exchange=exchange)
        shard.install(prod)
        with he.installed(), torch.no_grad():
            local = prod(shard.local_slice(x).half(), 301, txt.half()).sample
            full = shard.gather_frames(local).float()
            moved = shard.bytes_gathered
            FrameShard.uninstall(prod)
            out = None
            if rank == 0:
                ref = ora(x, torch.tensor(301), txt).sample
                half = prod(x[:, :, :T // 2].half(), 301, txt.half()).sample.float()
                whole = prod(x.half(), 301, txt.half()).sample.float()
                out = (rel_l2(full, ref), rel_l2(half, ref[:, :, :T // 2]), rel_l2(full, whole), moved)
        dist.barrier()
        dist.destroy_process_group()
        q.put(('ok', rank, out))
    except Exception:  # pragma: no cover
        q.put(('error', rank, traceback.format_exc()))


@pytest.mark.parametrize('exchange', ['kv', 'sites', 'auto'])
def test_frame_sharded_unet_host_mirror(exchange):
    """The real UNet, two ranks of 4 frames each (gloo): gathered output = full-clip oracle, and equal (to fp16 rounding
    noise) to the unsharded product on the full clip; without the exchange the half clip differs visibly."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get() for _ in range(world)]
    for p in procs:
        p.join(120)
    errs = [r for r in results if r[0] != 'ok']
    assert not errs, errs[0][2]
    err, uncoupled, vs_whole, moved = [r[2] for r in results if r[1] == 0][0]
    print(f'exchange={exchange}: sharded vs oracle {err:.3e}, vs unsharded product {vs_whole:.3e}, '
          f'uncoupled half clip {uncoupled:.3e}; {moved / 1e6:.2f} MB received per rank')
    assert err < 4e-3 and vs_whole < 2e-3
    assert uncoupled > 3 * err
    assert moved > 0
