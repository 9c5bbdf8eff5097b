#!/bin/bash
# Round 4: B pieces of a slab issued from a per-CU starting point (pp_sched 32) against the common order, every persistent shape
# (record of an experiment: in the build of that call — commit e46953b's working tree — pp_sched bit 32 switched the B piece
# rotation ON; since 6701f83 the rotation is the default and bit 32 switches it off)
TAG=${1:-r04j}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python tools/gemm_ab.py --scheds 0,32 --batch 2 > $O/${TAG}_brot_ab_b2.txt 2>&1
grep -v "^# .*differing" $O/${TAG}_brot_ab_b2.txt | cut -c1-150
