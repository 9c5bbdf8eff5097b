#!/bin/bash
# Round 4, final build: the whole -m gpu suite, traffic of THIS digest, bench line with both baselines, kernel trace, the
# frame-sharded bench on one rank (smoke), training bench.
TAG=${1:-r04fin2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 -rf ) > $O/${TAG}_pytest.log 2>&1
tail -n 26 $O/${TAG}_pytest.log | cut -c1-200
el pytest
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
cp $O/${TAG}_pmc_shape/gemm_hbm_traffic.json $R/profiles/gemm_hbm_traffic.json 2>/dev/null
el pmc_by_shape
timeout 500 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-2600
el bench
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r04 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
head -n 14 $O/${TAG}_kernel_stats.txt | cut -c1-170
el kernel_trace
timeout 300 python bench.py --config 4 --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline > $O/${TAG}_bench_cfg4_smoke.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg4_smoke.log | cut -c1-1200
el bench_cfg4_smoke
timeout 300 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg3.log | cut -c1-300
timeout 300 python tools/train_bench.py --steps 3 > $O/${TAG}_train_bench.txt 2>&1
tail -n 2 $O/${TAG}_train_bench.txt | cut -c1-250
el rest
