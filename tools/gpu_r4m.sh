#!/bin/bash
# Round 4: piece rotation, position-balanced (the variant order rotates round by round): B rotation on (default) / off (32),
# A rotation (64), under the product's dispatch, B = 2
# (record: in the build of that call s0 = B rotation only, 32 = none, 64 = A + B, 96 = A only; 6701f83 shipped A everywhere + B in the plain GEMMs)
TAG=${1:-r04m}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/gemm_ab.py --auto-scheds 0,32,64,96 --batch 2 --rounds 10 > $O/${TAG}_rot_ab_b2.txt 2>&1
grep -v "^# .*differing" $O/${TAG}_rot_ab_b2.txt | cut -c1-170
