#!/bin/bash
# Last call of round 2: the bench line of record (eager launch, CPU baseline included), then the full-width graph probe.
TAG=${1:-r02f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 270 python bench.py --steps 3 --warmup 1 > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-400
timeout 80 python tools/graph_probe.py --full > $O/${TAG}_graph_probe_full.log 2>&1
tail -n 12 $O/${TAG}_graph_probe_full.log | cut -c1-200
