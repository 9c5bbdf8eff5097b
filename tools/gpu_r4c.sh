#!/bin/bash
# Round 4, GPU call 3: flash-attention backward (kernel tests, autograd, training tests, training bench), shared CFG prefix,
# query blocks per wave by rule; bench.
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_autograd.py tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_training.py -m gpu -x -q -k "attention or gradient or shared_cfg or graph or inversion or train or step" --durations=8 -rf ) > $O/${TAG}_pytest_a.log 2>&1
tail -n 16 $O/${TAG}_pytest_a.log | cut -c1-220
timeout 300 python tools/train_bench.py --steps 3 > $O/${TAG}_train_bench.txt 2>&1
tail -n 5 $O/${TAG}_train_bench.txt | cut -c1-250
timeout 300 python tools/attn_ab.py --vars 1,2 > $O/${TAG}_attn_ab.txt 2>&1
head -n 7 $O/${TAG}_attn_ab.txt | cut -c1-200
for v in 1 0; do
  VSX_CFG_SHARED_PREFIX=$v timeout 400 python bench.py --no-cpu-baseline --steps 2 > $O/${TAG}_bench_shared$v.log 2>&1
  tail -n 1 $O/${TAG}_bench_shared$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'])"
done
