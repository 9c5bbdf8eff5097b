"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: clip sharding / timing reduction, and the two exchange
steps of the frame-sharded long-clip mode checked against single-process math (GroupNorm over all frames from
all-gathered partial sums; temporal attention from all-gathered K/V with global positional encoding)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(fn, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    results = [q.get() for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    errs = [r for r in results if isinstance(r, str)]
    assert not errs, errs
    return results


def _entry(fn, rank, world, port, q):
    import traceback
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    try:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        q.put(fn(rank, world))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        q.put(traceback.format_exc())
        raise


def _clip_parallel(rank, world):
    from videoswap_amd.distributed import max_over_ranks, shard_clips
    mine = shard_clips(7, rank, world)
    slow = max_over_ranks(1.0 + rank)
    return (mine, slow)


def test_clip_parallel_bookkeeping():
    res = _run(_clip_parallel)
    clips = sorted(c for mine, _ in res for c in mine)
    assert clips == list(range(7))                       # every clip exactly once, no collective on the data path
    assert all(abs(t - 2.0) < 1e-9 for _, t in res)      # whole-job time = slowest rank


def _frame_shard(rank, world):
    import torch.nn.functional as F
    from videoswap_amd.distributed import FrameShard
    torch.manual_seed(0)                                  # same full tensors on every rank
    B, T, HW, C, G = 2, 8, 6, 32, 8
    shard = FrameShard(T)
    x = torch.randn(B, T, HW, C)
    mine = x[:, shard.frame_offset:shard.frame_offset + shard.local_frames]
    # --- 5-D GroupNorm statistics from all-gathered partial sums (what vsx_groupnorm_stats/apply exchange) ---
    xs = mine.reshape(B, -1, G, C // G)
    partial = torch.stack([xs.sum((1, 3)), (xs * xs).sum((1, 3))], -1)[:, None]            # [B, 1 chunk, G, 2]
    allp = shard.gn_hook(partial.contiguous())
    assert allp.shape == (B, world, G, 2)
    tot = allp.sum(1)
    n = T * HW * (C // G)
    mean = tot[..., 0] / n
    var = tot[..., 1] / n - mean * mean
    ref = F.group_norm(x.reshape(B, T * HW, C).transpose(1, 2), G).transpose(1, 2).reshape(B, T, HW, C)
    got = ((mine.reshape(B, -1, G, C // G) - mean[:, None, :, None]) / torch.sqrt(var[:, None, :, None] + 1e-5))
    gn_err = (got.reshape(mine.shape) - ref[:, shard.frame_offset:shard.frame_offset + shard.local_frames]).abs().max()
    # --- temporal attention over ALL frames from all-gathered K/V ---
    q, k, v = torch.randn(B, T, HW, C), torch.randn(B, T, HW, C), torch.randn(B, T, HW, C)
    sl = slice(shard.frame_offset, shard.frame_offset + shard.local_frames)
    kv_local = torch.cat([k[:, sl], v[:, sl]], dim=-1).reshape(-1, 2 * C).contiguous()     # K | V columns
    kv_all, fk = shard.kv_gather_finish(shard.kv_gather_start(kv_local, B, shard.local_frames, HW))
    kg, vg = kv_all[:, :C], kv_all[:, C:]
    assert fk == T and torch.equal(kg.reshape(B, T, HW, C), k) and torch.equal(vg.reshape(B, T, HW, C), v)
    assert shard.bytes_gathered == (world - 1) * B * shard.local_frames * HW * 2 * C * 4

    def attn(qq, kk, vv):     # [B, f, HW, C] attention across the frame axis at every site
        s = torch.einsum('bfsc,bgsc->bsfg', qq, kk) / C ** 0.5
        return torch.einsum('bsfg,bgsc->bfsc', s.softmax(-1), vv)
    at_err = (attn(q[:, sl], kg.reshape(B, T, HW, C), vg.reshape(B, T, HW, C)) - attn(q, k, v)[:, sl]).abs().max()
    lat = torch.randn(1, 4, T, 3, 3)
    full = shard.gather_frames(shard.local_slice(lat))
    return (float(gn_err), float(at_err), bool(torch.equal(full, lat)), shard.frame_offset)


def test_frame_shard_exchange_steps():
    res = _run(_frame_shard)
    for gn_err, at_err, ok, off in res:
        assert gn_err < 1e-4 and at_err < 1e-5 and ok
    assert sorted(r[3] for r in res) == [0, 4]


def _site_shard(rank, world):
    """exchange='sites': frames -> sites -> frames is the identity, the site layout holds every frame of this rank's
    sites in global frame order, and temporal attention computed on it equals the full-clip result."""
    from videoswap_amd.distributed import FrameShard
    torch.manual_seed(0)
    B, T, HW, C = 2, 8, 3 * world, 16
    shard = FrameShard(T, exchange='sites')
    x = torch.randn(B, T, HW, C).half()
    sl = slice(shard.frame_offset, shard.frame_offset + shard.local_frames)
    mine = x[:, sl].reshape(-1, C).contiguous()
    hl = HW // world
    ys = shard.to_sites(mine, B, HW)
    want = x[:, :, rank * hl:(rank + 1) * hl].reshape(-1, C)            # all frames, my sites
    layout_ok = torch.equal(ys, want)
    moved = shard.bytes_gathered
    back = shard.to_frames(ys, B, HW)
    round_trip = torch.equal(back, mine)
    # temporal attention on the site layout == the full computation restricted to my sites
    q, k, v = (torch.randn(B, T, HW, C) for _ in range(3))

    def attn(qq, kk, vv):
        s = torch.einsum('bfsc,bgsc->bsfg', qq, kk) / C ** 0.5
        return torch.einsum('bsfg,bgsc->bfsc', s.softmax(-1), vv)

    def sites(t):
        return shard.to_sites(t[:, sl].reshape(-1, C).contiguous(), B, HW).view(B, T, hl, C)
    o_sites = attn(sites(q), sites(k), sites(v))                         # [B, T, hl, C]
    o_frames = shard.to_frames(o_sites.reshape(-1, C).contiguous(), B, HW).view(B, shard.local_frames, HW, C)
    at_err = (o_frames - attn(q, k, v)[:, sl]).abs().max()
    try:
        FrameShard(T, exchange='sites').sites_per_rank(7)
        refused = False
    except ValueError:
        refused = True
    return (layout_ok, round_trip, float(at_err), moved, refused)


@pytest.mark.parametrize('world', [2, 4])
def test_site_reshard_exchange(world):
    res = _run(_site_shard, world)
    B, T, HW, C = 2, 8, 3 * world, 16
    for layout_ok, round_trip, at_err, moved, refused in res:
        assert layout_ok and round_trip and refused
        assert at_err < 1e-5
        # one all-to-all moves (world-1)/world of the LOCAL activation: fp16, B x T/world frames x HW/world sites x C
        assert moved == (world - 1) * B * (T // world) * (HW // world) * C * 2


def _simulate_alltoall(sends, nouter, ninner, block, send_strides, recv_strides, out_numel):
    """vsx_alltoall_f16's contract (csrc/comm.cpp) for all ranks in one process: block (o, i) for peer p
    leaves rank r at p*ss[0] + o*ss[1] + i*ss[2] and lands on rank p at r*rs[0] + o*rs[1] + i*rs[2]."""
    world = len(sends)
    outs = [torch.full((out_numel,), float('nan')) for _ in range(world)]
    for r in range(world):
        for p in range(world):
            for o in range(nouter):
                for i in range(ninner):
                    so = p * send_strides[0] + o * send_strides[1] + i * send_strides[2]
                    do = r * recv_strides[0] + o * recv_strides[1] + i * recv_strides[2]
                    outs[p][do:do + block] = sends[r][so:so + block]
    return outs


@pytest.mark.parametrize('world', [2, 4])
def test_strided_alltoall_layouts_of_the_site_reshard(world):
    """The stride triples FrameShard hands to the C-ABI all-to-all (no pack / unpack copies) against the layouts the
    torch.distributed path produces: frames -> sites and back."""
    torch.manual_seed(1)
    B, T, HW, C = 2, 2 * world, 3 * world, 5
    f, hl = T // world, HW // world
    blk = hl * C
    x = torch.randn(B, T, HW, C)
    local = [x[:, r * f:(r + 1) * f].reshape(-1) for r in range(world)]                       # [b, f, (p, hl), c]
    from videoswap_amd.distributed import FrameShard
    to_sites = FrameShard.reshard_strides(world, f, blk)        # what to_sites / to_frames hand to the C ABI
    to_frames = to_sites[::-1]
    sites = _simulate_alltoall(local, B, f, blk, *to_sites, out_numel=B * T * hl * C)
    for r in range(world):
        want = x[:, :, r * hl:(r + 1) * hl].reshape(-1)                                       # all frames, my sites
        assert torch.equal(sites[r], want), r
    back = _simulate_alltoall(sites, B, f, blk, *to_frames, out_numel=B * f * HW * C)
    for r in range(world):
        assert torch.equal(back[r], local[r]), r


def _grad_allreduce(rank, world):
    """What DDP does for the adapter's 1.1 M parameters in the training step — mean of the gradients over the ranks —
    and what a GradScaler must do on several ranks: ONE skip / step decision.  Step 1: rank 1 alone overflows, and one
    parameter has no gradient on rank 0 only (fixed parameter list); every rank must skip and halve its scale without
    hanging.  Step 2: finite everywhere -> the mean gradient reaches the optimizer on every rank."""
    from videoswap_amd.trainer import VideoSwapTrainer
    torch.manual_seed(3)
    adapter = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))])
    base = [torch.randn(5, 3), torch.randn(7)]
    tr = object.__new__(VideoSwapTrainer)
    tr.adapter, tr.accelerator, tr.lr_scheduler = adapter, None, None
    tr.optimizer = torch.optim.SGD(adapter.parameters(), lr=1.0)
    tr.loss_scale, tr.growth_interval, tr._clean_steps, tr.skipped_steps = 4.0, 2000, 0, 0

    def loss_with(grads):       # a scalar whose gradient with respect to parameter i is grads[i] (None: not in the graph)
        return sum((p * g).sum() for p, g in zip(adapter, grads) if g is not None)
    bad = [base[0] * float('inf') if rank == 1 else base[0], None if rank == 0 else base[1]]
    applied = tr.backward_and_update(loss_with(bad))
    ok = (not applied) and tr.loss_scale == 2.0 and tr.skipped_steps == 1 and all(float(p.abs().sum()) == 0 for p in adapter)
    applied = tr.backward_and_update(loss_with([b * (rank + 1) for b in base]))
    mean = sum(range(1, world + 1)) / world
    return ok and applied and all(torch.allclose(p.detach(), -b * mean, atol=1e-6) for p, b in zip(adapter, base))


def test_adapter_gradient_allreduce():
    assert all(_run(_grad_allreduce))


# ---- the C-ABI collectives (csrc/comm.cpp) for P > 1 ranks without a second GPU: recording communicator + replay ----------
OP_SEND, OP_RECV, OP_ALLGATHER, OP_ALLREDUCE, OP_COPY, OP_GSTART, OP_GEND = 1, 2, 3, 4, 5, 6, 7


def _record_rank(lib, rank, world, call):
    """Rank `rank` of `world` on a recording communicator (include/vsx.h, ABI 8): `call(lib)` runs C-ABI collectives with
    FAKE buffer addresses (never dereferenced); returns the drained log as a list of 6-tuples."""
    import ctypes
    from videoswap_amd import _lib
    _lib.check(lib.vsx_comm_init_recording(rank, world), 'vsx_comm_init_recording')
    try:
        assert lib.vsx_comm_size() == world and lib.vsx_comm_rank() == rank
        call(lib)
        n = lib.vsx_comm_recorded(None, 0)
        buf = (ctypes.c_int64 * (6 * n))()
        assert lib.vsx_comm_recorded(buf, n) == n and lib.vsx_comm_recorded(None, 0) == 0
        return [tuple(buf[6 * i:6 * i + 6]) for i in range(n)]
    finally:
        _lib.check(lib.vsx_comm_destroy(), 'vsx_comm_destroy')
    assert lib.vsx_comm_size() == 0


def _replay(logs, sends, recvs):
    """Execute the ranks' logs against each other on host tensors.  NCCL point-to-point matching: the k-th send of rank a
    to rank b meets the k-th receive of rank b from rank a; element counts must agree; every send / receive must be
    inside ONE group per rank (a deadlock-free all-to-all); all-gathers must be issued with the same count in the same
    order on every rank."""
    world = len(logs)
    for r, log in enumerate(logs):
        depth, groups = 0, 0
        for op, peer, so, do, n, es in log:
            if op == OP_GSTART:
                depth += 1
                groups += 1
            elif op == OP_GEND:
                depth -= 1
            elif op in (OP_SEND, OP_RECV):
                assert depth == 1, f'rank {r}: point-to-point call outside a group'
                assert 0 <= peer < world and peer != r
        assert depth == 0 and groups <= 1, f'rank {r}: {groups} groups'
    for a in range(world):
        for op, peer, so, do, n, es in logs[a]:
            if op == OP_COPY:
                recvs[a].view(-1)[do:do + n] = sends[a].view(-1)[so:so + n]
        for b in range(world):
            if a == b:
                continue
            s = [rec for rec in logs[a] if rec[0] == OP_SEND and rec[1] == b]
            d = [rec for rec in logs[b] if rec[0] == OP_RECV and rec[1] == a]
            assert len(s) == len(d), f'{len(s)} sends {a}->{b} but {len(d)} receives'
            for (_, _, so, _, n, es), (_, _, _, do, n2, es2) in zip(s, d):
                assert n == n2 and es == es2 == sends[a].element_size()
                recvs[b].view(-1)[do:do + n] = sends[a].view(-1)[so:so + n]
    gathers = [[rec for rec in log if rec[0] == OP_ALLGATHER] for log in logs]
    assert len({len(g) for g in gathers}) == 1
    for k in range(len(gathers[0])):
        assert len({gathers[r][k][4] for r in range(world)}) == 1, 'all-gather counts differ across ranks'
        for r in range(world):
            _, _, _, do, n, _ = gathers[r][k]
            for q in range(world):
                so_q = gathers[q][k][2]
                recvs[r].view(-1)[do + q * n:do + (q + 1) * n] = sends[q].view(-1)[so_q:so_q + n]


def _have_lib():
    from videoswap_amd import _lib
    return os.path.exists(_lib.LIB_PATH)


@pytest.mark.parametrize('world', [2, 4, 8])
def test_c_abi_alltoall_schedule_replayed_across_ranks(world):
    """vsx_alltoall_f16 as FrameShard(backend='rccl', exchange='sites') calls it (FrameShard._alltoall_strided with
    FrameShard.reshard_strides), rank by rank on recording communicators, replayed: frames -> sites must equal the torch
    path of `to_sites` (pack, all_to_all_single, unpack: tests above) = the plain re-layout of the full clip, and
    sites -> frames must bring every rank's frames back bit for bit."""
    if not _have_lib():
        pytest.skip('libvsx.so not built')
    import ctypes
    from videoswap_amd import _lib
    from videoswap_amd.distributed import FrameShard
    lib = _lib.load()
    b, f, hw, c = 2, 3, 8 * world, 5            # f local frames per rank, hw sites (hw / world per rank), c channels
    hl = hw // world
    blk = hl * c
    g = torch.Generator().manual_seed(40 + world)
    full = torch.randn(b, world * f, hw, c, generator=g).half()              # the whole clip [B, F_total, hw, C]
    local = [full[:, r * f:(r + 1) * f].contiguous() for r in range(world)]   # rank r: its frames, all sites
    arr = ctypes.c_int64 * 3
    FAKE_SRC, FAKE_DST = 0x10000000, 0x20000000

    def a2a(send_st, recv_st):
        return lambda lib_: _lib.check(lib_.vsx_alltoall_f16(ctypes.c_void_p(FAKE_SRC), ctypes.c_void_p(FAKE_DST), b, f, blk,
                                                             arr(*send_st), arr(*recv_st), None), 'vsx_alltoall_f16')
    send_st, recv_st = FrameShard.reshard_strides(world, f, blk)
    logs = [_record_rank(lib, r, world, a2a(send_st, recv_st)) for r in range(world)]
    # one group, (P - 1) * b * f sends and as many receives per rank, b * f local copies: nothing is packed or staged
    for log in logs:
        assert sum(rec[0] == OP_SEND for rec in log) == (world - 1) * b * f
        assert sum(rec[0] == OP_RECV for rec in log) == (world - 1) * b * f
        assert sum(rec[0] == OP_COPY for rec in log) == b * f
        assert all(rec[4] == blk for rec in log if rec[0] in (OP_SEND, OP_RECV, OP_COPY))
    sites = [torch.full((b, world * f, hl, c), float('nan')).half() for _ in range(world)]
    _replay(logs, local, sites)
    for r in range(world):                      # rank r now holds sites [r*hl, (r+1)*hl) of EVERY frame, frames in global order
        assert torch.equal(sites[r], full[:, :, r * hl:(r + 1) * hl])
    # and back: the same call with the two stride triples exchanged
    logs = [_record_rank(lib, r, world, a2a(recv_st, send_st)) for r in range(world)]
    back = [torch.full((b, f, hw, c), float('nan')).half() for _ in range(world)]
    _replay(logs, sites, back)
    for r in range(world):
        assert torch.equal(back[r], local[r])


@pytest.mark.parametrize('world', [2, 4])
def test_c_abi_allgathers_replayed_across_ranks(world):
    """vsx_allgather_kv (one ncclAllGather per batch item inside one group, straight into the [B, P, f*hw, 2C] buffer the
    temporal attention kernel reads as [B, F_total, hw, 2C]) and vsx_allgather_f32 (GroupNorm partial sums in rank order),
    recorded per rank and replayed."""
    if not _have_lib():
        pytest.skip('libvsx.so not built')
    import ctypes
    from videoswap_amd import _lib
    lib = _lib.load()
    b, f, hw, c2 = 2, 3, 4, 6
    per_batch = f * hw * c2
    g = torch.Generator().manual_seed(50 + world)
    full = torch.randn(b, world * f, hw, c2, generator=g).half()             # K|V of all frames
    local = [full[:, r * f:(r + 1) * f].contiguous() for r in range(world)]
    FAKE_SRC, FAKE_DST = 0x10000000, 0x20000000

    def kv(lib_):
        _lib.check(lib_.vsx_allgather_kv(ctypes.c_void_p(FAKE_SRC), ctypes.c_void_p(FAKE_DST), b, per_batch, None),
                   'vsx_allgather_kv')
    logs = [_record_rank(lib, r, world, kv) for r in range(world)]
    for log in logs:
        assert [rec[0] for rec in log] == [OP_GSTART] + [OP_ALLGATHER] * b + [OP_GEND]
    out = [torch.full((b, world * f, hw, c2), float('nan')).half() for _ in range(world)]
    _replay(logs, local, out)
    for r in range(world):
        assert torch.equal(out[r], full)        # every rank: all frames in global order, per batch item

    part = [torch.randn(b, 2, 4, 2, generator=g) for _ in range(world)]      # [nimg, nchunks, groups, 2] fp32
    n = part[0].numel()

    def f32(lib_):
        _lib.check(lib_.vsx_allgather_f32(ctypes.c_void_p(FAKE_SRC), ctypes.c_void_p(FAKE_DST), n, None), 'vsx_allgather_f32')
    logs = [_record_rank(lib, r, world, f32) for r in range(world)]
    allp = [torch.full((world, b, 2, 4, 2), float('nan')) for _ in range(world)]
    _replay(logs, part, allp)
    for r in range(world):
        assert torch.equal(allp[r], torch.stack(part))


@pytest.mark.parametrize('config,cps_arg', [(2, None), (2, 4), (4, None)])
def test_bench_plumbing_two_ranks_gloo(config, cps_arg, tmp_path):
    """bench.py's own multi-rank code — process-group init, barriers, max-over-ranks, the whole-job metric arithmetic and the
    ONE JSON line from rank 0 — launched the way the driver launches it (torch.distributed.run, 2 ranks), on CPU with gloo and
    the clip loop stubbed (VSX_BENCH_STUB_CLIP=1: no model, no kernels).  The driver's first SCALE run must not be the first
    execution of these lines; the same branches run on one MI355X rank with the real loop and the
    nccl backend in tools/gpu_steps.sh dist1."""
    import json
    import subprocess
    import sys
    from util import ROOT
    env = dict(os.environ, VSX_BENCH_STUB_CLIP='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--config', str(config), '--no-cpu-baseline']
    if cps_arg is not None:
        cmd += ['--clips-per-step', str(cps_arg)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'rank 0 prints ONE JSON line, got {len(lines)}'
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['higher_is_better'] is True
    frames = 64 if config == 4 else 16
    cps = out['config']['latents'][0]                # clips denoised together in one step (1 for the long clip)
    # default: ONE clip per GPU per step (SURVEY.md §8(d) Config 5 = one configs[1] clip per GPU); --clips-per-step 4: the batch
    assert cps == (cps_arg or 1)
    assert out['config']['stub'] is True and out['metric'].startswith('STUBBED')     # a stubbed line says so
    assert 'throughput_mode' not in out              # the second leg is a single-GPU measurement
    clips_job = 3 if config == 4 else 2 * 3 * cps    # the ranks share one long clip per step / every rank its own clips
    assert out['scaling'] == ('strong' if config == 4 else 'weak')
    assert abs(out['value'] - clips_job * frames / (out['ms_per_step'] * 3 / 1e3)) <= 0.02 * out['value']
    assert out['ms_per_step'] >= 10.0                # the stub sleeps 10 ms per clip: the timed region really ran 3 steps
    assert 'roofline' not in out and out['vs_baseline'] is None


@pytest.mark.parametrize('cps', [None, 1, 2, 4])
def test_bench_single_process_metric_arithmetic(cps, tmp_path):
    """The default launch (`python bench.py`, one process, no torchrun) with the clip loop stubbed: one JSON line, `value` =
    clips per step x steps x 16 frames / the timed region; the default is ONE clip per step = SURVEY.md §8(d) Config 2
    (latents [1,4,16,64,64], what the reference's test.py drives), a batch is named as such in `metric` and `workload`."""
    import json
    import subprocess
    import sys
    from util import ROOT
    env = dict(os.environ, VSX_BENCH_STUB_CLIP='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'VSX_FORCE_DISTRIBUTED'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '0', '--no-cpu-baseline']
    if cps is not None:
        cmd += ['--clips-per-step', str(cps)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    n = 1 if cps is None else cps
    assert out['config']['latents'] == [n, 4, 16, 64, 64] and out['n_gpus'] == 1 and out['steps'] == 3
    assert '@ 50 DDIM steps' in out['metric'] and out['config']['stub'] is True
    if n == 1:
        assert 'BASELINE.json configs[1]: 16-frame 512x512 clip' in out['config']['workload'] and 'together' not in out['metric']
    else:
        assert f'{n} independent BASELINE.json configs[1] clips batched' in out['config']['workload']
        assert f'{n} independent clips denoised together' in out['metric']
    assert 'throughput_mode' not in out              # (stubbed / no GPU: the second leg does not run)
    assert f'inversion (B={n})' in out['config']['workload'] and f'sampling (B={2 * n})' in out['config']['workload']
    assert abs(out['value'] - n * 3 * 16 / (out['ms_per_step'] * 3 / 1e3)) <= 0.02 * out['value']
    assert out['unit'] == 'frames/s' and out['dtype'] == 'f16' and out['data'] == 'synthetic' and out['scaling'] == 'weak'
