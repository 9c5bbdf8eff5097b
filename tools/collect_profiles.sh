#!/bin/bash
# Copies the judged summaries of a tools/gpu_validate.sh (= tools/gpu_steps.sh) call from gpurun_out/ (scratch) into profiles/
# (tracked).      bash tools/collect_profiles.sh r05v r05
TAG=${1:?validation tag}; RN=${2:?round prefix, e.g. r05}
O=gpurun_out; P=profiles
cpl() { [ -f "$1" ] && tail -n 1 "$1" > "$2"; }
cpl $O/${TAG}_bench.txt $P/${RN}_bench.json
cpl $O/${TAG}_benchcfg3.txt $P/${RN}_bench_cfg3.json
cpl $O/${TAG}_bench448.txt $P/${RN}_bench_448x768.json
cpl $O/${TAG}_bench2.txt $P/${RN}_bench_two_clips_per_step.json
cpl $O/${TAG}_dist1_clip.txt $P/${RN}_bench_forced_distributed_one_rank.json
cpl $O/${TAG}_dist1_long.txt $P/${RN}_bench_cfg4_forced_distributed_one_rank.json
for f in kernel_stats train; do [ -f $O/${TAG}_$f.txt ] && cp $O/${TAG}_$f.txt $P/${RN}_$f.txt; done
[ -f $O/${TAG}_pmc_sq/summary.txt ] && cp $O/${TAG}_pmc_sq/summary.txt $P/${RN}_pmc_sq.txt
for B in 1 2 4; do
  [ -f $O/${TAG}_gemm_traffic_by_shape_b$B.txt ] && cp $O/${TAG}_gemm_traffic_by_shape_b$B.txt $P/${RN}_gemm_traffic_by_shape_b$B.txt
  [ -f $O/${TAG}_gemm_hbm_traffic_b$B.json ] && python tools/pmc_by_shape.py --merge $P/gemm_hbm_traffic.json $O/${TAG}_gemm_hbm_traffic_b$B.json
  [ -f $O/${TAG}_kernel_stats_b$B.txt ] && cp $O/${TAG}_kernel_stats_b$B.txt $P/${RN}_kernel_stats_b$B.txt
done
[ -f $O/${TAG}_pytest.txt ] && ( echo "# python -m pytest tests -m gpu -x -q --durations=12 -rf   (tools/gpu_validate.sh $TAG)"; grep -v "^$" $O/${TAG}_pytest.txt | tail -n 40 ) > $P/${RN}_gpu_tests.txt
for f in parity_fullwidth.json parity_cfg3.json; do [ -f $O/$f ] && cp $O/$f $P/${RN}_$f; done
python - <<PY
import json
from videoswap_amd.build import source_digest
t = json.load(open('$P/gemm_hbm_traffic.json'))
for k, v in sorted(t.items()):
    print('traffic entry', k, 'matches the tree:', v['lib_digest'] == source_digest(), v['lib_digest'][:12], 'ratio %.3f' % v['ratio'])
PY
