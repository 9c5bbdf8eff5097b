#!/bin/bash
# The GPU tests tools/gpu_small.sh does not run, then the three bench lines (default one with both baseline legs).
TAG=${1:-rest}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 520 python -m pytest tests -m gpu -q -rf --durations=8 --ignore=tests/test_kernels_gpu.py --ignore=tests/test_unet_gpu.py --ignore=tests/test_frame_shard_gpu.py --ignore=tests/test_processors_gpu.py --ignore=tests/test_cfg3_fullwidth_gpu.py ) > $O/${TAG}_pytest.log 2>&1
grep -v "^$" $O/${TAG}_pytest.log | tail -n 16 | cut -c1-200
timeout 300 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-1200
timeout 200 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg3.log | cut -c100-230
timeout 200 python bench.py --latent-h 56 --latent-w 96 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_448x768.log 2>&1
tail -n 1 $O/${TAG}_bench_448x768.log | cut -c100-230
