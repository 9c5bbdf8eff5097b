#!/bin/bash
# A/B of two library builds on ONE box: per-shape GEMM table (tools/gemm_ab.py), bench line, kernel stats of the loop.
# The other build is a development variant of videoswap_amd/build.py (VSX_LIB_VARIANT=<name>, lib/libvsx_<name>.so).
TAG=${1:-epi}
OTHER=${2:-prev}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf ) > $O/${TAG}_kernel_tests.log 2>&1
tail -n 6 $O/${TAG}_kernel_tests.log | cut -c1-220
for v in $OTHER ""; do
  n=${v:-new}
  VSX_LIB_VARIANT=$v timeout 300 python tools/gemm_ab.py --batch 2 --rounds 3 > $O/${TAG}_gemm_ab_$n.txt 2>&1
  echo "== gemm_ab $n"; tail -n 3 $O/${TAG}_gemm_ab_$n.txt | cut -c1-200
done
for v in $OTHER "" $OTHER ""; do
  n=${v:-new}
  VSX_LIB_VARIANT=$v timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_$n.log 2>&1
  echo "bench $n: $(tail -n 1 $O/${TAG}_bench_$n.log | cut -c100-230)"
done
for v in $OTHER ""; do
  n=${v:-new}
  ( cd /tmp && export TMPDIR=/tmp && VSX_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_$n -o t -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof_$n.log 2>&1 )
  DB=$(find $O/${TAG}_prof_$n -name '*.db' | head -n 1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_$n.txt 2>&1
  find $O/${TAG}_prof_$n -type f -size +4M -delete 2>/dev/null
  echo "---- $n"; head -n 12 $O/${TAG}_kernel_stats_$n.txt | cut -c1-150
done
( timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_fullwidth_gpu.py -m gpu -q -rf -k "not sequential and not 24_frames and not host_oracle" ) > $O/${TAG}_model_tests.log 2>&1
tail -n 5 $O/${TAG}_model_tests.log | cut -c1-220
