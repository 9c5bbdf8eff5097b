#!/bin/bash
# Round 4: small-M GEMMs of the 16x16 / 8x8 levels under every configuration the library can be forced into
TAG=${1:-r04n}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 500 python tools/small_m_sweep.py > $O/${TAG}_small_m_sweep.txt 2>&1
cat $O/${TAG}_small_m_sweep.txt | cut -c1-200
