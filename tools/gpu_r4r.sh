#!/bin/bash
# Round 4: the convolutions with about one 128x320 tile per CU on the persistent kernel's 128-row form (shared A slab) against
# the tile kernels they run on today, B = 1 and B = 2
TAG=${1:-r04r}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for b in 1 2; do
  timeout 300 python tools/gemm_ab.py --kinds conv --variants auto:1:0:0,pp128:3:0:0,pp256:2:0:0 --batch $b --rounds 8 > $O/${TAG}_conv_pp128_b$b.txt 2>&1
  grep -v "^# .*differing" $O/${TAG}_conv_pp128_b$b.txt | cut -c1-150
done
