#!/bin/bash
# Round 4: producer row statistics with v_dot2 sums — test + alternating bench
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "row_statistics or persistent_linear" 2>&1 | tail -n 3 )
for v in 1 0 1 0; do
  VSX_ROW_STATS_PRODUCER=$v timeout 400 python bench.py --no-cpu-baseline --steps 1 > $O/${TAG}_bench_rs$v.log 2>&1
  tail -n 1 $O/${TAG}_bench_rs$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('producer stats $v:', d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'])"
done
