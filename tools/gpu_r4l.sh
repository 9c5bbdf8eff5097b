#!/bin/bash
# Round 4: piece rotation — A pieces of the persistent kernel (pp_sched 64), A and B pieces of the tile kernels (128), both (192)
# under the product's dispatch, every shape of a forward, B = 2 and B = 1
# (record: bits 64 / 128 existed only in the build of that call — A piece rotation as an opt-in, rotation in the tile kernels;
# 6701f83 made the A rotation the default and dropped the tile-kernel one)
TAG=${1:-r04l}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for b in 2 1; do
  timeout 400 python tools/gemm_ab.py --auto-scheds 0,64,128,192 --batch $b > $O/${TAG}_rot_ab_b$b.txt 2>&1
  grep -v "^# .*differing" $O/${TAG}_rot_ab_b$b.txt | cut -c1-170
done
