#!/bin/bash
# Round 4, GPU call 4: where a flash-attention wave spends a key tile (timing build: per-segment cycle totals).
TAG=${1:-r04d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export VSX_SKIP_DIGEST_CHECK=1
{
for qb in 1 2; do
  echo "== VSX_ATTN_QB=$qb"
  VSX_ATTN_QB=$qb timeout 120 python tools/gemm_timing.py attn 32 8 4096 4096 40
  VSX_ATTN_QB=$qb timeout 120 python tools/gemm_timing.py attn 2 8 4096 4096 40
  VSX_ATTN_QB=$qb timeout 120 python tools/gemm_timing.py attn 32 8 1024 1024 80
done
} > $O/${TAG}_attn_timing.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_attn_timing.txt | cut -c1-200
