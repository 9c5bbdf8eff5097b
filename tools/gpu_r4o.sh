#!/bin/bash
# Round 4: the new dispatch (four ring slots for <= 256 workgroups of 128x160) against the forced two-slot tile, both batches;
# persistent-conv equality tests once more on this build
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "persistent or conv or linear or gemm" 2>&1 | tail -n 3 )
for b in 1 2; do
  timeout 400 python tools/gemm_ab.py --variants auto:1:0:0,t128x160:0:0:2,t128x160d:0:0:18 --batch $b --rounds 8 > $O/${TAG}_deep_ab_b$b.txt 2>&1
  grep -v "^# .*differing" $O/${TAG}_deep_ab_b$b.txt | cut -c1-150
done
