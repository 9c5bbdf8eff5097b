#!/bin/bash
# Round 4: shared A slab (default) / private A slab (16) x common B piece order / per-CU rotation (32), one process
# (record: build of e46953b's working tree, where bit 32 = B piece rotation ON and 16 = private A slab; today bit 32 = common order)
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python tools/gemm_ab.py --kinds conv --scheds 0,16,32,48 --batch 2 --rounds 7 > $O/${TAG}_conv_ab4_b2.txt 2>&1
grep -v "^# .*differing" $O/${TAG}_conv_ab4_b2.txt | cut -c1-170
