#!/bin/bash
# kernel trace of an arbitrary python command on the GPU box:  gpurun -- 'bash tools/gpu_prof_cmd.sh <tag> <script> [args]'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o t -- python $R/"$@" > $O/${TAG}.log 2>&1 )
DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
tail -n 5 $O/${TAG}.log | cut -c1-200
head -n 40 $O/${TAG}_kernel_stats.txt | cut -c1-190
