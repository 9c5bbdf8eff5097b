#!/bin/bash
# Round 3, GPU call 2: configs[2] at full width (parity against the reference-controller golden, bench line, kernel
# trace), training / all-to-all entry points now in libvsx.so, 20+20-step loop parity with timings, 448x768 bench,
# default bench with both baseline legs, fixed MFMA/VALU overlap probe, RCCL two-ranks-one-GPU probe.
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.log 2>&1 || { cat $O/${TAG}_lib.log; exit 3; }
( timeout 400 python -m pytest tests/test_cfg3_fullwidth_gpu.py -m gpu -q -s -rf ) > $O/${TAG}_cfg3_test.log 2>&1
grep -E "cfg3_T4|passed|failed|Error|assert" $O/${TAG}_cfg3_test.log | cut -c1-700 | tail -n 8
el cfg3_test
( timeout 300 python -m pytest tests/test_autograd.py tests/test_training.py tests/test_frame_shard_gpu.py -m gpu -q -rf -k "not long_clip" ) > $O/${TAG}_promoted.log 2>&1
tail -n 3 $O/${TAG}_promoted.log | cut -c1-200
el promoted
timeout 300 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.log 2>&1
tail -n 1 $O/${TAG}_bench_cfg3.log | cut -c1-1500
el bench_cfg3
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_cfg3 -o r03 -- python $R/bench.py --config 3 --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --prof-samples 0 > $O/${TAG}_prof_cfg3.log 2>&1 )
DB=$(find $O/${TAG}_prof_cfg3 -name '*.db' | head -n 1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_cfg3.txt 2>&1
find $O/${TAG}_prof_cfg3 -type f -size +4M -delete 2>/dev/null
head -n 30 $O/${TAG}_kernel_stats_cfg3.txt | cut -c1-180; tail -n 1 $O/${TAG}_kernel_stats_cfg3.txt
el prof_cfg3
timeout 300 python bench.py --latent-h 56 --latent-w 96 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_448x768.log 2>&1
tail -n 1 $O/${TAG}_bench_448x768.log | cut -c1-1200
el bench_448x768
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
tail -n 1 $O/${TAG}_bench.log | cut -c1-2500
el bench_default
( timeout 500 python -m pytest tests/test_fullwidth_gpu.py -m gpu -q -s -rf -k "sequential" ) > $O/${TAG}_loops.log 2>&1
grep -E "loops_|passed|failed" $O/${TAG}_loops.log | cut -c1-700
el loops
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu mfma_valu.hip && timeout 60 /tmp/mfma_valu ) > $O/${TAG}_mfma_valu.txt 2>&1
tail -n 4 $O/${TAG}_mfma_valu.txt | cut -c1-300
el mfma_valu
( timeout 120 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 2 tools/rccl_same_gpu_probe.py ) > $O/${TAG}_rccl_same_gpu.txt 2>&1
grep -E "^rank" $O/${TAG}_rccl_same_gpu.txt | cut -c1-400
el rccl_probe
