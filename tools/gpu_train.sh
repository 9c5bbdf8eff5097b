#!/bin/bash
# training-step checks on the GPU box: gradient tests, trainer tests, one timed step at the reference's configuration
TAG=${1:-train}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_autograd.py tests/test_training.py -m gpu -q -rf ) > $O/${TAG}_tests.log 2>&1
tail -n 3 $O/${TAG}_tests.log | cut -c1-200
timeout 300 python tools/train_bench.py --frames 16 --latent 64 --steps 3 > $O/${TAG}_bench.txt 2>&1
tail -n 4 $O/${TAG}_bench.txt | cut -c1-250
