#!/bin/bash
# Round 4: sub-pixel form of the nearest-2x convolutions (ABI 9): kernel + UNet + full-width parity tests, per-shape traffic of
# this digest, bench with the form on (default) and off
TAG=${1:-r04s}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -x -q 2>&1 | tail -n 4 )
( timeout 500 python -m pytest tests/test_fullwidth_gpu.py tests/test_cfg3_fullwidth_gpu.py -m gpu -x -q 2>&1 | tail -n 4 )
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sub-pixel on :', d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'], d['roofline']['traffic_ratio'])"
VSX_CONV_SUBPIXEL=0 timeout 300 python bench.py --no-cpu-baseline > $O/${TAG}_bench_off.log 2>&1
tail -n 1 $O/${TAG}_bench_off.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sub-pixel off:', d['value'], d['readings']['inversion_s_per_clip'], d['readings']['sampling_s_per_clip'], d['roofline']['frac'])"
