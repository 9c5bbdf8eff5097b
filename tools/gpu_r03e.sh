#!/bin/bash
# A/B of the persistent kernel's option bits at the UNet's shapes (B = 2 and B = 1).
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/gemm_ab.py --batch 2 --rounds 3 --scheds ${2:-0,4,8,12,16,32} > $O/${TAG}_opts_b2.txt 2>&1
cat $O/${TAG}_opts_b2.txt | cut -c1-220
timeout 300 python tools/gemm_ab.py --batch 1 --rounds 3 --scheds ${2:-0,4,8,12,16,32} > $O/${TAG}_opts_b1.txt 2>&1
tail -n 42 $O/${TAG}_opts_b1.txt | cut -c1-220
