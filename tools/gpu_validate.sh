#!/bin/bash
# Validation call of a round: the full -m gpu suite, the default bench (both baseline legs), the configs[2] and 448x768
# bench lines, kernel trace (10 + 10 steps) with gap analysis, SQ counter passes, per-shape HBM traffic.
#   gpurun --timeout 2400 -- 'bash tools/gpu_validate.sh r03v'
TAG=${1:-r04v}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 -rf ) > $O/${TAG}_pytest.log 2>&1
tail -n 24 $O/${TAG}_pytest.log | cut -c1-220
el pytest
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
cp $O/${TAG}_pmc_shape/gemm_hbm_traffic.json $R/profiles/gemm_hbm_traffic.json 2>/dev/null    # bench reads it (digest-keyed)
el pmc_by_shape
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-3000
el bench
timeout 300 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg3.log | cut -c1-400
timeout 300 python bench.py --latent-h 56 --latent-w 96 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_448x768.log 2>&1
tail -n 1 $O/${TAG}_bench_448x768.log | cut -c1-400
el bench_other
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r04 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
head -n 14 $O/${TAG}_kernel_stats.txt | cut -c1-170; tail -n 3 $O/${TAG}_kernel_stats.txt | cut -c1-200
el kernel_trace
bash tools/pmc_sq.sh ${TAG}_pmc_sq
cat $O/${TAG}_pmc_sq/passes.txt; head -n 20 $O/${TAG}_pmc_sq/summary.txt | cut -c1-200
el pmc_sq
# conv slab order "taps of a channel slab back to back" (pp_sched bit 4): its per-shape traffic, for the record
VSX_PP_SCHED=4 bash tools/pmc_by_shape.sh ${TAG}_pmc_shape_sched4 > $O/${TAG}_pmc_shape_sched4.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape_sched4.txt
el pmc_by_shape_sched4
timeout 300 python tools/train_bench.py --steps 3 > $O/${TAG}_train_bench.txt 2>&1
tail -n 3 $O/${TAG}_train_bench.txt | cut -c1-250
el train_bench
