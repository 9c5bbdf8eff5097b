#!/bin/bash
# run one python tool on the GPU box, output into gpurun_out/<tag>.txt:  gpurun -- 'bash tools/gpu_small.sh <tag> <script> [args]'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python "$@" > gpurun_out/${TAG}.txt 2>&1
tail -n 40 gpurun_out/${TAG}.txt | cut -c1-220
