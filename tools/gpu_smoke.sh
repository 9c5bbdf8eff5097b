#!/bin/bash
# what the driver runs at round end, minus the test suite: __graft_entry__.smoke() and the default bench line
TAG=${1:-smoke}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'], d['roofline']['traffic_ratio'], d['cpu_baseline']['value'])"
