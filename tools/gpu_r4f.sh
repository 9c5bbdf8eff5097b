#!/bin/bash
# Round 4: GEGLU epilogue staged in fp16 — timing build (epilogue cycles per tile) + per-shape A/B of the GEGLU launches + tests
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
VSX_SKIP_DIGEST_CHECK=1 timeout 300 python tools/gemm_timing.py pp > $O/${TAG}_pp_timing.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_pp_timing.txt | cut -c1-200
timeout 300 python tools/gemm_ab.py --kinds geglu --batch 2 > $O/${TAG}_gemm_ab_geglu_b2.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_gemm_ab_geglu_b2.txt | cut -c1-200
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "geglu or persistent" 2>&1 | tail -n 3 )
