#!/bin/bash
# Round 4: LayerNorm row statistics from the producing GEMM's epilogue — kernel test, A/B of the bench loop, kernel stats
TAG=${1:-r04g}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -x -q -k "row_statistics or persistent or layer_norm or golden or shared_cfg" 2>&1 | tail -n 4 )
for v in 1 0 1 0; do
  VSX_ROW_STATS_PRODUCER=$v timeout 400 python bench.py --no-cpu-baseline --steps 1 > $O/${TAG}_bench_rs$v.log 2>&1
  tail -n 1 $O/${TAG}_bench_rs$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('producer stats $v:', d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'])"
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r04 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
grep -i "row_stats\|gemm_pp_kernel<[12], false, [0189]>\|kernel time" $O/${TAG}_kernel_stats.txt | cut -c1-170
