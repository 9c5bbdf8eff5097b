#!/bin/bash
# Round 3, GPU call 1: everything that has never executed on an MI355X (development library: piece schedules, packed B,
# backward kernels, strided all-to-all; exchange='sites'), a same-box bench of the shipped library, the loop-timing probe
#  and the MFMA/VALU overlap micro-benchmark.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r03a.sh r03a'
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
el lib
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-400
el bench
( VSX_TEST_SITES=1 timeout 300 python -m pytest tests/test_frame_shard_gpu.py -m gpu -q -s -rf ) > $O/${TAG}_pytest_sites.log 2>&1
grep -E "exchange=|passed|failed" $O/${TAG}_pytest_sites.log | cut -c1-250
el sites
# ---- development library ----
VSX_LIB_VARIANT=next python -c "from videoswap_amd import _lib; l=_lib.load(); print('next lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_next_lib.log 2>&1 || { cat $O/${TAG}_next_lib.log; exit 0; }
for s in 0 3 4 5 6; do
  ( VSX_LIB_VARIANT=next VSX_TEST_PP_SCHED=$s timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf ) > $O/${TAG}_next_pp_s$s.log 2>&1
  echo "next lib, pp_sched $s: $(tail -n 1 $O/${TAG}_next_pp_s$s.log | cut -c1-120)"
done
el pp_tests
( VSX_LIB_VARIANT=next timeout 100 python -m pytest tests/test_frame_shard_gpu.py -q -k alltoall -rf ) > $O/${TAG}_next_alltoall.log 2>&1
echo "next lib, alltoall: $(tail -n 1 $O/${TAG}_next_alltoall.log | cut -c1-120)"
# gradient path / training step on the development library's backward kernels (never run on hardware before)
( VSX_LIB_VARIANT=next timeout 400 python -m pytest tests/test_autograd.py tests/test_training.py -m gpu -q -s -rf ) > $O/${TAG}_next_training.log 2>&1
grep -E "loss:|worst cosine|level|passed|failed|Error" $O/${TAG}_next_training.log | cut -c1-200 | tail -n 30
el training_tests
VSX_LIB_VARIANT=next timeout 240 python tools/gemm_ab.py --batch 2 --rounds 3 --scheds 0,3,4,5,6 > $O/${TAG}_next_sched_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_sched_b2.txt | cut -c1-250
el sched_b2
VSX_LIB_VARIANT=next timeout 240 python tools/gemm_ab.py --batch 2 --rounds 3 --scheds 0,4,16,20 --bpack > $O/${TAG}_next_bpack_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_bpack_b2.txt | cut -c1-250
el bpack_b2
VSX_LIB_VARIANT=next timeout 240 python tools/gemm_ab.py --batch 2 --rounds 2 --scheds 0,32,64 > $O/${TAG}_next_diag_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_diag_b2.txt | cut -c1-250
el diag_b2
VSX_LIB_VARIANT=next timeout 200 python tools/gemm_ab.py --batch 1 --rounds 3 --scheds 0,3,4 > $O/${TAG}_next_sched_b1.txt 2>&1
tail -n 2 $O/${TAG}_next_sched_b1.txt | cut -c1-250
el sched_b1
VSX_LIB_VARIANT=next timeout 300 python tools/train_bench.py --frames 16 --latent 64 --steps 2 > $O/${TAG}_next_train_bench.txt 2>&1
tail -n 4 $O/${TAG}_next_train_bench.txt | cut -c1-250
el train_bench
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu mfma_valu.hip && timeout 60 /tmp/mfma_valu ) > $O/${TAG}_mfma_valu.txt 2>&1
tail -n 6 $O/${TAG}_mfma_valu.txt | cut -c1-250
el mfma_valu
timeout 300 python tools/loop_timing_probe.py --steps 10 --out $O/${TAG}_loop_timing_probe.json > $O/${TAG}_loop_timing_probe.txt 2>&1
tail -n 12 $O/${TAG}_loop_timing_probe.txt | cut -c1-300
el loop_probe
