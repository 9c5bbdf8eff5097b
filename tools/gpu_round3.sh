#!/bin/bash
# gpurun call 3: persistent-kernel gate, A/B (tile / pp256 / pp128 / auto), benches (eager tile, eager pp, graphs), full suite
TAG=${1:-r02c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf > $O/${TAG}_pp_tests.log 2>&1
PP_RC=$?
tail -n 4 $O/${TAG}_pp_tests.log
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k "temporal or big or benchmark_shape" -rf > $O/${TAG}_kernel_tests.log 2>&1
tail -n 6 $O/${TAG}_kernel_tests.log | cut -c1-300
if [ $PP_RC -ne 0 ]; then
  echo "persistent kernel tests rc=$PP_RC: rest of the call runs with VSX_GEMM_PP=0"
  export VSX_GEMM_PP=0
else
  timeout 300 python tools/gemm_ab.py --batch 2 --rounds 4 > $O/${TAG}_gemm_ab_b2.txt 2>&1
  timeout 300 python tools/gemm_ab.py --batch 1 --rounds 4 > $O/${TAG}_gemm_ab_b1.txt 2>&1
  tail -n 2 $O/${TAG}_gemm_ab_b2.txt $O/${TAG}_gemm_ab_b1.txt
fi
VSX_GEMM_PP=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs > $O/${TAG}_bench_tile_eager.log 2>&1
tail -n 1 $O/${TAG}_bench_tile_eager.log | cut -c1-200
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs > $O/${TAG}_bench_pp_eager.log 2>&1
tail -n 1 $O/${TAG}_bench_pp_eager.log | cut -c1-200
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_pp_graphs.log 2>&1
tail -n 1 $O/${TAG}_bench_pp_graphs.log | cut -c1-200
( time timeout 900 python -m pytest tests -m gpu -q --durations=25 -rf -x --deselect tests/test_kernels_gpu.py ) > $O/${TAG}_pytest.log 2>&1
tail -n 40 $O/${TAG}_pytest.log | cut -c1-300
