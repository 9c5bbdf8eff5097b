#!/bin/bash
# Small validation after a kernel change: kernel tests, UNet / frame-shard / processor tests, per-shape HBM traffic (the
# digest-keyed JSON bench.py reads), kernel stats of 10 + 10 steps, two bench lines.
TAG=${1:-small}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf ) > $O/${TAG}_kernel_tests.log 2>&1
tail -n 5 $O/${TAG}_kernel_tests.log | cut -c1-220
( timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_frame_shard_gpu.py tests/test_processors_gpu.py tests/test_cfg3_fullwidth_gpu.py -m gpu -q -rf -x ) > $O/${TAG}_model_tests.log 2>&1
grep -E "passed|failed|error" $O/${TAG}_model_tests.log | tail -n 3 | cut -c1-220
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o t -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
head -n 16 $O/${TAG}_kernel_stats.txt | cut -c1-150
for k in 1 2; do
  timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
  echo "bench: $(tail -n 1 $O/${TAG}_bench.log | cut -c100-230)"
done
