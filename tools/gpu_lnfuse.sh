#!/bin/bash
# LayerNorm folded into the consuming GEMMs (VSX_LN_FUSE=1, the default) against the standalone kernel (=0): bench lines
# (alternating, one box) and the loop's kernel time
TAG=${1:-lnfuse}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for f in 0 1 0 1; do
  VSX_LN_FUSE=$f timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_f$f.log 2>&1
  echo "VSX_LN_FUSE=$f: $(tail -n 1 $O/${TAG}_bench_f$f.log | cut -c100-230)"
done
for f in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && VSX_LN_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_f$f -o t -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof_f$f.log 2>&1 )
  DB=$(find $O/${TAG}_prof_f$f -name '*.db' | head -n 1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_f$f.txt 2>&1
  find $O/${TAG}_prof_f$f -type f -size +4M -delete 2>/dev/null
  echo "---- VSX_LN_FUSE=$f"; head -n 1 $O/${TAG}_kernel_stats_f$f.txt; grep -E "layernorm|row_stats|gemm_pp_kernel|gemm_kernel<128, 160|gemm_kernel<256" $O/${TAG}_kernel_stats_f$f.txt | cut -c1-150
done
