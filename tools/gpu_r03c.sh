#!/bin/bash
# Round 3, GPU call 3: micro-benchmarks (MFMA / VALU overlap with per-role loops; GEMM main-loop structures) and the
# per-shape HBM traffic of vsx_gemm_f16.
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu mfma_valu.hip && timeout 60 /tmp/mfma_valu ) > $O/${TAG}_mfma_valu.txt 2>&1
tail -n 4 $O/${TAG}_mfma_valu.txt | cut -c1-400
el mfma_valu
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_loop gemm_loop.hip && timeout 120 /tmp/gemm_loop 256 2560 160 && timeout 120 /tmp/gemm_loop 256 320 40 && timeout 60 /tmp/gemm_loop 8 2560 160 ) > $O/${TAG}_gemm_loop.txt 2>&1
cat $O/${TAG}_gemm_loop.txt | cut -c1-200
el gemm_loop
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape
el pmc_by_shape
