#!/bin/bash
# Round 4: persistent GEMM, K loop vs epilogue cycles per tile (timing build)
TAG=${1:-r04e}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export VSX_SKIP_DIGEST_CHECK=1
timeout 300 python tools/gemm_timing.py pp > $O/${TAG}_pp_timing.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_pp_timing.txt | cut -c1-200
