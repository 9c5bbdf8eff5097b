#!/bin/bash
# Round 4, GPU call 1: VALU issue costs at 1/2/4 waves per SIMD, flash-attention variants A/B (peeled partial tile is in every
# variant), convolution slab order A/B (pp_sched 0 vs 4), the tests of everything touched so far, one bench line.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r4a.sh r04a'
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip && timeout 120 /tmp/valu_rate ) > $O/${TAG}_ubench_valu_rate.txt 2>&1
cat $O/${TAG}_ubench_valu_rate.txt | cut -c1-220
el valu_rate
timeout 300 python tools/attn_ab.py > $O/${TAG}_attn_ab.txt 2>&1
cat $O/${TAG}_attn_ab.txt | cut -c1-260
el attn_ab
timeout 400 python tools/gemm_ab.py --scheds 0,4 --kinds conv --batch 2 > $O/${TAG}_gemm_conv_order_ab_b2.txt 2>&1
cat $O/${TAG}_gemm_conv_order_ab_b2.txt | cut -c1-200
el gemm_ab
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_frame_shard_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q --durations=8 -rf ) > $O/${TAG}_pytest_a.log 2>&1
tail -n 16 $O/${TAG}_pytest_a.log | cut -c1-220
el pytest_a
( time timeout 900 python -m pytest tests/test_fullwidth_gpu.py -m gpu -x -q --durations=12 -rf ) > $O/${TAG}_pytest_b.log 2>&1
tail -n 20 $O/${TAG}_pytest_b.log | cut -c1-220
el pytest_b
timeout 400 python bench.py --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-1500
el bench
VSX_PP_SCHED=4 timeout 400 python bench.py --no-cpu-baseline --steps 1 > $O/${TAG}_bench_sched4.log 2>&1
tail -n 1 $O/${TAG}_bench_sched4.log | cut -c1-600
el bench_sched4
