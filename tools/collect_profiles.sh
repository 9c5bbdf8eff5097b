#!/bin/bash
# Copies the judged summaries of a tools/gpu_validate.sh call from gpurun_out/ (scratch) into profiles/ (tracked).
#   bash tools/collect_profiles.sh r03w r03
TAG=${1:?validation tag}; RN=${2:?round prefix, e.g. r03}
O=gpurun_out; P=profiles
tail -n 1 $O/${TAG}_bench.log > $P/${RN}_bench.json
tail -n 1 $O/${TAG}_bench_cfg3.log > $P/${RN}_bench_cfg3.json
tail -n 1 $O/${TAG}_bench_448x768.log > $P/${RN}_bench_448x768.json
cp $O/${TAG}_kernel_stats.txt $P/${RN}_kernel_stats.txt
cp $O/${TAG}_pmc_sq/summary.txt $P/${RN}_pmc_sq.txt
cp $O/${TAG}_pmc_shape/by_shape.txt $P/${RN}_gemm_traffic_by_shape.txt
cp $O/${TAG}_pmc_shape/gemm_hbm_traffic.json $P/gemm_hbm_traffic.json
( echo "# python -m pytest tests -m gpu -x -q --durations=12 -rf   (tools/gpu_validate.sh $TAG)"; grep -v "^$" $O/${TAG}_pytest.log | tail -n 40 ) > $P/${RN}_gpu_tests.txt
for f in parity_fullwidth.json parity_cfg3.json; do [ -f $O/$f ] && cp $O/$f $P/${RN}_$f; done
python - <<PY
import json
from videoswap_amd.build import source_digest
t = json.load(open('$P/gemm_hbm_traffic.json'))
print('traffic digest matches the tree:', t['lib_digest'] == source_digest(), t['lib_digest'][:12])
PY
