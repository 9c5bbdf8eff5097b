#!/bin/bash
# tile kernels vs 256-row / 128-row persistent tiles vs the automatic choice at the UNet's shapes
TAG=${1:-r03k}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/gemm_ab.py --batch 2 --rounds 4 > $O/${TAG}_gemm_ab_b2.txt 2>&1
timeout 300 python tools/gemm_ab.py --batch 1 --rounds 4 > $O/${TAG}_gemm_ab_b1.txt 2>&1
grep -v "differing" $O/${TAG}_gemm_ab_b2.txt | cut -c1-190
grep -v "differing" $O/${TAG}_gemm_ab_b1.txt | cut -c1-190
