#!/bin/bash
# Round 4, final build (taps-inner convolution order by default): kernel tests, the loop parity, traffic of THIS digest, bench
# line with both baselines, kernel trace, the CPU leg at the full T = 16.
TAG=${1:-r04fin}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_frame_shard_gpu.py -m gpu -x -q --durations=6 -rf ) > $O/${TAG}_pytest_a.log 2>&1
tail -n 12 $O/${TAG}_pytest_a.log | cut -c1-200
el pytest_a
( time timeout 900 python -m pytest tests/test_fullwidth_gpu.py tests/test_cfg3_fullwidth_gpu.py -m gpu -x -q -k "benchmark_shape or sequential_steps_full_width or T16 or outlier" --durations=6 -rf ) > $O/${TAG}_pytest_b.log 2>&1
tail -n 10 $O/${TAG}_pytest_b.log | cut -c1-200
el pytest_b
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
cp $O/${TAG}_pmc_shape/gemm_hbm_traffic.json $R/profiles/gemm_hbm_traffic.json 2>/dev/null
el pmc_by_shape
timeout 500 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-2600
el bench
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r04 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
head -n 12 $O/${TAG}_kernel_stats.txt | cut -c1-170
el kernel_trace
timeout 900 python tools/cpu_baseline_T16.py $O/${TAG}_cpu_baseline_T16.json > $O/${TAG}_cpu_baseline_T16.log 2>&1
cat $O/${TAG}_cpu_baseline_T16.json
el cpu_baseline_T16
