#!/bin/bash
# GEMM main-loop micro-benchmark sweeps: how the ping-pong loop's matrix-pipe duty depends on the number of active CUs
# and on which operand panels are shared (L2 hits) or private (fabric / HBM).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_loop gemm_loop.hip || exit 1
{
for n in 256; do /tmp/gemm_loop $n 2560 160 1; done
echo "---- small panels (K = 320: A 164 KB per workgroup) ----"
for n in 256; do /tmp/gemm_loop $n 320 160 1; done
} > $O/${1:-r03f}_gemm_loop_sweep.txt 2>&1
grep -v "^status" $O/${1:-r03f}_gemm_loop_sweep.txt | cut -c1-200
