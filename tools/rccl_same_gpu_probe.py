#!/usr/bin/env python
"""Can two RCCL ranks share ONE GPU?  (a way to run the frame-shard collectives across real ranks
without a multi-GPU node.)  Launch:  python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 2 \
tools/rccl_same_gpu_probe.py   — both ranks use cuda:0; prints what RCCL answers."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
        x = torch.full((1024,), float(rank + 1), device='cuda')
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(f'rank {rank}: all_reduce over two ranks on one GPU -> {float(x[0])} (expected 3.0)', flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f'rank {rank}: RCCL refuses two ranks on one device: {type(e).__name__}: {str(e)[:300]}', flush=True)


if __name__ == '__main__':
    main()
