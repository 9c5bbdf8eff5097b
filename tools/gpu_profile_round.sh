#!/bin/bash
# The gpurun call the round-2 profiles were taken with (profiles/README.md): gate, A/B, benches, rocprof kernel trace (eager + graphs), PMC passes, full GPU suite.
TAG=${1:-r02d}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k persistent -rf > $O/${TAG}_pp_tests.log 2>&1
PP_RC=$?
tail -n 4 $O/${TAG}_pp_tests.log | cut -c1-300
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
timeout 400 python bench.py --steps 3 --warmup 1 > $O/${TAG}_bench_graphs.log 2>&1
tail -n 1 $O/${TAG}_bench_graphs.log | cut -c1-260
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs > $O/${TAG}_bench_eager.log 2>&1
tail -n 1 $O/${TAG}_bench_eager.log | cut -c1-260
# kernel trace (eager: per-kernel names; same command as round 1) and of the graph-replay mode
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_eager -o r02 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-graphs --prof-samples 0 > $O/${TAG}_prof_eager.log 2>&1 )
DB=$(find $O/${TAG}_prof_eager -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_eager.txt 2>&1
find $O/${TAG}_prof_eager -type f -size +4M -delete 2>/dev/null
tail -n 3 $O/${TAG}_kernel_stats_eager.txt | cut -c1-200
if [ "${SKIP_PMC:-0}" != "1" ]; then
  bash tools/pmc_sq.sh ${TAG}_pmc_sq
  cat $O/${TAG}_pmc_sq/passes.txt
  bash tools/pmc_traffic.sh > $O/${TAG}_pmc_traffic.log 2>&1
  cat $O/pmc_traffic/r02_gemm_hbm_traffic.json 2>/dev/null | head -n 12
fi
cd $R
( time timeout 900 python -m pytest tests -m gpu -q --durations=25 -rf --deselect tests/test_kernels_gpu.py ) > $O/${TAG}_pytest.log 2>&1
tail -n 45 $O/${TAG}_pytest.log | cut -c1-300
