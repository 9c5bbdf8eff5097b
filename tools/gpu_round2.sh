#!/bin/bash
# gpurun call: persistent-kernel smoke first (bounded), then the A/B table, the full GPU suite and two short benches.
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf > $O/${TAG}_pp_tests.log 2>&1
PP_RC=$?
tail -4 $O/${TAG}_pp_tests.log
if [ $PP_RC -ne 0 ]; then
  echo "persistent kernel tests rc=$PP_RC: rest of the call runs with VSX_GEMM_PP=0"
  export VSX_GEMM_PP=0
else
  timeout 400 python tools/gemm_ab.py --batch 2 > $O/${TAG}_gemm_ab_b2.txt 2>&1
  timeout 400 python tools/gemm_ab.py --batch 1 > $O/${TAG}_gemm_ab_b1.txt 2>&1
  tail -3 $O/${TAG}_gemm_ab_b2.txt $O/${TAG}_gemm_ab_b1.txt
fi
( time timeout 1000 python -m pytest tests -m gpu -q --durations=15 -rf ) > $O/${TAG}_pytest.log 2>&1
tail -6 $O/${TAG}_pytest.log
VSX_GEMM_PP=0 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_tile.log 2>&1
tail -1 $O/${TAG}_bench_tile.log | cut -c1-400
if [ $PP_RC -eq 0 ]; then
  timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_pp.log 2>&1
  tail -1 $O/${TAG}_bench_pp.log | cut -c1-400
fi
