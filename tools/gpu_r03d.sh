#!/bin/bash
# Round 3, GPU call 4: A/B of the persistent kernel's new option bits (pp_sched + 4 conv slab order, + 8 2-D tile walk,
# + 16 / + 32 priority variants) at the UNet's shapes; MFMA/VALU probe with priorities.
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu mfma_valu.hip && timeout 60 /tmp/mfma_valu ) > $O/${TAG}_mfma_valu.txt 2>&1
tail -n 6 $O/${TAG}_mfma_valu.txt | cut -c1-400
el mfma_valu
( timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf ) > $O/${TAG}_pp_tests.log 2>&1
tail -n 2 $O/${TAG}_pp_tests.log | cut -c1-200
for s in 4 8 32; do
  ( VSX_TEST_PP_SCHED=$s timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf ) > $O/${TAG}_pp_tests_s$s.log 2>&1
  echo "pp_sched $s: $(tail -n 1 $O/${TAG}_pp_tests_s$s.log | cut -c1-160)"
done
el pp_tests
timeout 300 python tools/gemm_ab.py --batch 2 --rounds 3 --scheds 0,4,8,12,16,32 > $O/${TAG}_opts_b2.txt 2>&1
cat $O/${TAG}_opts_b2.txt | cut -c1-200
el opts_b2
timeout 300 python tools/gemm_ab.py --batch 1 --rounds 3 --scheds 0,4,8,12,16,32 > $O/${TAG}_opts_b1.txt 2>&1
tail -n 42 $O/${TAG}_opts_b1.txt | cut -c1-200
el opts_b1
