#!/bin/bash
# One gpurun call: the GPU test suite, the counter passes and a short bench.  Usage (from the build container):
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 1100 python -m pytest tests -m gpu -q --durations=20 -rf ) > $O/${TAG}_pytest.log 2>&1
tail -5 $O/${TAG}_pytest.log
if [ "${SKIP_PMC:-0}" != "1" ]; then
  rocprofv3 -L > $O/${TAG}_counters_list.txt 2>&1
  bash tools/pmc_sq.sh ${TAG}_pmc_sq
  cat $O/${TAG}_pmc_sq/passes.txt
fi
cd $R
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
tail -2 $O/${TAG}_bench.log
