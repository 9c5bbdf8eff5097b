#!/bin/bash
# A/B tables of the GEMM back ends at the UNet's shapes (B = 2 and B = 1) in one GPU call:
#   gpurun -- 'bash tools/gpu_gemm_ab.sh <tag>'            tile kernels vs 256- / 128-row persistent tiles vs the automatic choice
#   gpurun -- 'bash tools/gpu_gemm_ab.sh <tag> 0,8,1,2'    pp_sched variants of the 256-row persistent kernel (8 = linear tile walk)
TAG=${1:-gemm_ab}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
S=${2:+--scheds $2}
timeout 300 python tools/gemm_ab.py --batch 2 --rounds 4 $S > $O/${TAG}_b2.txt 2>&1
timeout 300 python tools/gemm_ab.py --batch 1 --rounds 4 $S > $O/${TAG}_b1.txt 2>&1
grep -v "differing" $O/${TAG}_b2.txt | cut -c1-200
grep -v "differing" $O/${TAG}_b1.txt | cut -c1-200
