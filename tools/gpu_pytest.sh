#!/bin/bash
# run a pytest selection on the GPU box:  gpurun -- 'bash tools/gpu_pytest.sh <tag> <pytest args>'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( time timeout 900 python -m pytest "$@" -m gpu -q -s -rf --durations=8 ) > gpurun_out/${TAG}.log 2>&1
grep -E "^unet_|^loops_|^cfg3|passed|failed|Error|assert|^[0-9.]+s " gpurun_out/${TAG}.log | cut -c1-400 | tail -n 30
