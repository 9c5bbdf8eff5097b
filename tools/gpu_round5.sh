#!/bin/bash
# Final short gpurun call of round 2: the test files the previous call did not reach, the graph diagnostic, final bench.
TAG=${1:-r02e}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
timeout 170 python bench.py --steps 3 --warmup 1 > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-300
( time timeout 200 python -m pytest tests/test_unet_gpu.py tests/test_vae.py tests/test_processors_gpu.py tests/test_pipeline_gpu.py tests/test_swap_flow_gpu.py -m gpu -q --durations=8 -rf ) > $O/${TAG}_pytest_rest.log 2>&1
tail -n 16 $O/${TAG}_pytest_rest.log | cut -c1-250
( time timeout 200 python -m pytest tests/test_kernels_gpu.py -q --durations=5 -rf ) > $O/${TAG}_pytest_kernels.log 2>&1
tail -n 14 $O/${TAG}_pytest_kernels.log | cut -c1-250
timeout 100 python tools/graph_probe.py > $O/${TAG}_graph_probe.log 2>&1
tail -n 12 $O/${TAG}_graph_probe.log | cut -c1-200
