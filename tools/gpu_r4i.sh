#!/bin/bash
# Round 4: shared A slab of the stride-1 3x3 convolutions (one A slab per filter row, read at three row offsets):
# equality test against the private-slab form, per-shape A/B (pp_sched 16 = private A slab per tap), B = 2 and B = 1
TAG=${1:-r04i}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "persistent_conv" 2>&1 | tail -n 3 )
timeout 300 python tools/gemm_ab.py --kinds conv --scheds 0,16 --batch 2 > $O/${TAG}_conv_ab_b2.txt 2>&1
cat $O/${TAG}_conv_ab_b2.txt | cut -c1-150
timeout 300 python tools/gemm_ab.py --kinds conv --scheds 0,16 --batch 1 > $O/${TAG}_conv_ab_b1.txt 2>&1
tail -n 1 $O/${TAG}_conv_ab_b1.txt
