#!/bin/bash
# Round 4, GPU call 2: flash attention with 64 queries per wave (attn_var 4 / 6), graph tests after the capture fix, bench A/B.
TAG=${1:-r04b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
timeout 300 python tools/attn_ab.py --vars 0,4,6,5 > $O/${TAG}_attn_ab.txt 2>&1
cat $O/${TAG}_attn_ab.txt | cut -c1-300
( time timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "graph or attention" --durations=5 -rf ) > $O/${TAG}_pytest_a.log 2>&1
tail -n 12 $O/${TAG}_pytest_a.log | cut -c1-220
for v in 0 4; do
  VSX_ATTN_VAR=$v timeout 400 python bench.py --no-cpu-baseline --steps 1 > $O/${TAG}_bench_var$v.log 2>&1
  tail -n 1 $O/${TAG}_bench_var$v.log | cut -c1-330
done
VSX_ATTN_VAR=4 timeout 300 python bench.py --no-cpu-baseline --steps 1 --graphs > $O/${TAG}_bench_graphs.log 2>&1
tail -n 1 $O/${TAG}_bench_graphs.log | cut -c1-330
