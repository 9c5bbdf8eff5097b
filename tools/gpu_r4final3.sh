#!/bin/bash
# Round 4, last touch (M*ldc check relaxed for the transposed store; combine kernel's stores): kernel tests, traffic of THIS
# digest, bench, the frame-sharded bench on one rank (T = 64 unsharded, 2 DDIM steps)
TAG=${1:-r04fin3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -x -q 2>&1 | tail -n 3 )
bash tools/pmc_by_shape.sh ${TAG}_pmc_shape > $O/${TAG}_pmc_shape.txt 2>&1
tail -n 1 $O/${TAG}_pmc_shape.txt
cp $O/${TAG}_pmc_shape/gemm_hbm_traffic.json $R/profiles/gemm_hbm_traffic.json 2>/dev/null
timeout 500 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-2600
timeout 400 python bench.py --config 4 --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline > $O/${TAG}_bench_cfg4_smoke.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg4_smoke.log | cut -c1-1500
