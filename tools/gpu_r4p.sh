#!/bin/bash
# Round 4: flash attention, key-tile rotation per 256-query block (option attn_rot) against the common sweep order
# (record: option attn_rot existed only in the build of that call; removed again in 4478709 — no effect)
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/attn_ab.py --option attn_rot --vars 0,1 --rounds 10 > $O/${TAG}_attn_rot_ab.txt 2>&1
cat $O/${TAG}_attn_rot_ab.txt | cut -c1-200
