#!/bin/bash
# First gpurun call of the next round: (1) the full -m gpu suite + bench on the shipped library (incl. the pieces that
# could not run on hardware in round 2: exchange='sites', chunked temporal attention), (2) the development library
# (videoswap_amd/csrc/experimental, VSX_LIB_VARIANT=next): persistent-kernel tests bit-for-bit under every candidate
# piece schedule, then the A/B tables of the schedules and of the packed-B experiment at the UNet's shapes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_next_round.sh r03a'
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( time timeout 600 python -m pytest tests -m gpu -x -q --durations=10 -rf ) > $O/${TAG}_pytest.log 2>&1
tail -n 8 $O/${TAG}_pytest.log | cut -c1-250
( time VSX_TEST_SITES=1 timeout 400 python -m pytest tests/test_frame_shard_gpu.py -m gpu -q -s -rf ) > $O/${TAG}_pytest_sites.log 2>&1
grep -E "exchange=|passed|failed" $O/${TAG}_pytest_sites.log | cut -c1-250
timeout 400 python bench.py --steps 2 --warmup 1 > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-600
# ---- development library ----
VSX_LIB_VARIANT=next python -c "from videoswap_amd import _lib; l=_lib.load(); print('next lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_next_lib.log 2>&1 || { cat $O/${TAG}_next_lib.log; exit 0; }
for s in 0 3 4 5 6; do
  ( VSX_LIB_VARIANT=next VSX_TEST_PP_SCHED=$s timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf ) > $O/${TAG}_next_pp_s$s.log 2>&1
  echo "next lib, pp_sched $s: $(tail -n 1 $O/${TAG}_next_pp_s$s.log | cut -c1-120)"
done
( VSX_LIB_VARIANT=next timeout 100 python -m pytest tests/test_frame_shard_gpu.py -q -k alltoall -rf ) > $O/${TAG}_next_alltoall.log 2>&1
echo "next lib, alltoall: $(tail -n 1 $O/${TAG}_next_alltoall.log | cut -c1-120)"
VSX_LIB_VARIANT=next timeout 300 python tools/gemm_ab.py --batch 2 --rounds 4 --scheds 0,3,4,5,6 > $O/${TAG}_next_sched_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_sched_b2.txt | cut -c1-250
VSX_LIB_VARIANT=next timeout 300 python tools/gemm_ab.py --batch 2 --rounds 4 --scheds 0,4,16,20 --bpack > $O/${TAG}_next_bpack_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_bpack_b2.txt | cut -c1-250
VSX_LIB_VARIANT=next timeout 300 python tools/gemm_ab.py --batch 2 --rounds 3 --scheds 0,4,32,36,64,68 > $O/${TAG}_next_diag_b2.txt 2>&1
tail -n 3 $O/${TAG}_next_diag_b2.txt | cut -c1-250
VSX_LIB_VARIANT=next timeout 300 python tools/gemm_ab.py --batch 1 --rounds 4 --scheds 0,3,4,5,6 > $O/${TAG}_next_sched_b1.txt 2>&1
tail -n 2 $O/${TAG}_next_sched_b1.txt | cut -c1-250
# ---- gradient path / training step on the development library's backward kernels ----
( VSX_LIB_VARIANT=next timeout 300 python -m pytest tests/test_autograd.py tests/test_training.py -m gpu -q -s -rf ) > $O/${TAG}_next_training.log 2>&1
grep -E "loss:|worst cosine|level|passed|failed" $O/${TAG}_next_training.log | cut -c1-200
VSX_LIB_VARIANT=next timeout 300 python tools/train_bench.py --frames 16 --latent 64 --steps 2 > $O/${TAG}_next_train_bench.txt 2>&1
tail -n 4 $O/${TAG}_next_train_bench.txt | cut -c1-250
# ---- do MFMA and VALU work of different waves overlap on a SIMD?  (decides whether a warp-specialised attention pays) ----
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu mfma_valu.hip && timeout 60 /tmp/mfma_valu ) > $O/${TAG}_mfma_valu.txt 2>&1
tail -n 4 $O/${TAG}_mfma_valu.txt | cut -c1-250
