#!/bin/bash
# LayerNorm folded into the consuming GEMMs: kernel tests, model parity, and an A/B of the loop's kernel time on one box
TAG=${1:-lnfuse}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "folded or row_stats or layer_norm or persistent_linear or geglu" ) > $O/${TAG}_kernel_tests.log 2>&1
tail -n 12 $O/${TAG}_kernel_tests.log | cut -c1-220
( timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_processors_gpu.py tests/test_fullwidth_gpu.py -m gpu -q -rf -k "not sequential and not 24_frames and not host_oracle" ) > $O/${TAG}_model_tests.log 2>&1
tail -n 8 $O/${TAG}_model_tests.log | cut -c1-220
for f in 0 1; do
  VSX_LN_FUSE=$f timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_f$f.log 2>&1
  echo "VSX_LN_FUSE=$f: $(tail -n 1 $O/${TAG}_bench_f$f.log | cut -c1-260)"
done
for f in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && VSX_LN_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_f$f -o t -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof_f$f.log 2>&1 )
  DB=$(find $O/${TAG}_prof_f$f -name '*.db' | head -n 1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_f$f.txt 2>&1
  find $O/${TAG}_prof_f$f -type f -size +4M -delete 2>/dev/null
  echo "---- VSX_LN_FUSE=$f"; head -n 1 $O/${TAG}_kernel_stats_f$f.txt; grep -E "layernorm|row_stats|gemm_pp_kernel<2, false|gemm_pp_kernel<1, false|gemm_kernel<128, 160|gemm_kernel<256" $O/${TAG}_kernel_stats_f$f.txt | cut -c1-150
done
