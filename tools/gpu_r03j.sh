#!/bin/bash
# kernel trace of the headline loop (10 + 10 steps) on the current build + gap analysis; full bench
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-900
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r03 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
head -n 28 $O/${TAG}_kernel_stats.txt | cut -c1-170; tail -n 12 $O/${TAG}_kernel_stats.txt | cut -c1-200
