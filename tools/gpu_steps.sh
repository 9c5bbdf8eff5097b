#!/bin/bash
# One parameterised GPU call: `gpurun --timeout N -- 'bash tools/gpu_steps.sh TAG step [step ...]'`.  Every step writes
# gpurun_out/TAG_<step>.txt (merged back by gpurun) and prints its tail; a failing step does not stop the following ones.
# Steps (round 5 folded the one-off gpu_r4*.sh scripts into this file):
#   lib            the library loads; prints its source digest
#   square         tools/gemm_square.py        calibration of the persistent GEMM on square problems
#   ksweep         tools/k_sweep.py            fixed cost against per-slab cost of the small-M GEMMs
#   xcdab          tools/gemm_ab.py            XCD block grid of the tile kernels on / off, 16x16 and 8x8 levels, B = 1 and 2
#   timing         tools/gemm_timing.py        phase stamps of the tile kernels on the small-M shapes (timing build)
#   ab:<args>      tools/gemm_ab.py <args>     free-form A/B ("," for spaces)
#   py:<args>      python <args>               any tool ("," for spaces)
#   hip:<name>     hipcc tools/ubench/<name>.hip && run it
#   kernels        pytest tests/test_kernels_gpu.py
#   pytest         the whole -m gpu suite
#   bench          python bench.py             (default line, both baseline legs)
#   benchq         python bench.py --no-cpu-baseline --steps 2 --warmup 1
#   benchcfg3      python bench.py --config 3 ...   (configs[2]: the full swap path)
#   bench448       python bench.py --latent-h 56 --latent-w 96 ...   (448x768 frames: 26 of the 30 reference option files)
#   bench2         python bench.py --clips-per-step 2 ...   (two clips denoised together)
#   smoke          __graft_entry__.smoke()
#   trace          rocprofv3 --kernel-trace --stats of a 10 + 10-step clip -> TAG_kernel_stats.txt
#   pmcshape[:N]   tools/pmc_by_shape.sh       per-shape fabric traffic at N clips per step (default 1); the entry b<N> is merged into
#                                              profiles/gemm_hbm_traffic.json on the box (later bench steps of the SAME call see it) and comes
#                                              back as gpurun_out/TAG_gemm_hbm_traffic_bN.json: merge it here with tools/pmc_by_shape.py --merge
#   pmcsq          tools/pmc_sq.sh             SQ counters (MFMA busy, LDS conflicts)
#   attn           tools/attn_ab.py
#   train          tools/train_bench.py
#   dist1          the distributed branches of bench.py on ONE rank (torchrun --nproc-per-node 1, VSX_FORCE_DISTRIBUTED=1)
TAG=${1:-r05}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
show() { tail -n ${2:-12} $O/${TAG}_$1.txt | cut -c1-${3:-260}; }
for STEP in "$@"; do
  case $STEP in
    lib)
      python -c "from videoswap_amd import _lib; l=_lib.load(); print('lib ok', l.vsx_source_digest().decode()[:12])" > $O/${TAG}_lib.txt 2>&1; show lib 3 ;;
    square)
      timeout 300 python tools/gemm_square.py > $O/${TAG}_square.txt 2>&1; show square 30 ;;
    ksweep)
      timeout 600 python tools/k_sweep.py > $O/${TAG}_ksweep.txt 2>&1; show ksweep 60 ;;
    xcdab)
      for B in 2 1; do
        timeout 400 python tools/gemm_ab.py --batch $B --levels 16,8 --rounds 7 --base auto:1:0:0:0 --variants grid:1:0:0:1 > $O/${TAG}_xcdab_b$B.txt 2>&1
        tail -n 40 $O/${TAG}_xcdab_b$B.txt | cut -c1-200
      done ;;
    timing)
      {
        for spec in "2 4096 1280 1280" "18 4096 1280 1280" "2 4096 1280 320" "18 4096 1280 5120" "6 1024 1280 1280" "1 4096 1280 1280" "18 8192 1280 1280"; do
          set -- $spec
          VSX_SKIP_DIGEST_CHECK=1 VSX_TIMING_RES=1 VSX_TUNE_TILE=$1 timeout 120 python tools/gemm_timing.py run $2 $3 $4
        done
      } > $O/${TAG}_timing.txt 2>&1; show timing 120 ;;
    ab:*)
      A=${STEP#ab:}; N=$(echo "$A" | md5sum | cut -c1-6)
      timeout 900 python tools/gemm_ab.py ${A//,/ } > $O/${TAG}_ab_$N.txt 2>&1; echo "# gemm_ab ${A//,/ }"; show ab_$N 70 ;;
    py:*)
      A=${STEP#py:}; N=$(echo "$A" | md5sum | cut -c1-6)
      timeout 900 python ${A//,/ } > $O/${TAG}_py_$N.txt 2>&1; echo "# python ${A//,/ }"; show py_$N 70 ;;
    hip:*)                    # hip:NAME — compile tools/ubench/NAME.hip for gfx950 on the box and run it
      A=${STEP#hip:}
      ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/$A tools/ubench/$A.hip && timeout 300 /tmp/$A ) > $O/${TAG}_hip_$A.txt 2>&1; show hip_$A 40 ;;
    kernels)
      ( time timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -rf ) > $O/${TAG}_kernels.txt 2>&1; show kernels 15 ;;
    pytest)
      ( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 -rf ) > $O/${TAG}_pytest.txt 2>&1; show pytest 24 ;;
    bench)
      timeout 500 python bench.py > $O/${TAG}_bench.txt 2>&1; show bench 1 3500 ;;
    benchq)
      timeout 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/${TAG}_benchq.txt 2>&1; show benchq 1 3500 ;;
    benchcfg3)
      timeout 400 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_benchcfg3.txt 2>&1; show benchcfg3 1 600 ;;
    bench448)
      timeout 400 python bench.py --latent-h 56 --latent-w 96 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench448.txt 2>&1; show bench448 1 600 ;;
    bench2)
      timeout 400 python bench.py --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-reading > $O/${TAG}_bench2.txt 2>&1; show bench2 1 1500 ;;
    benchenv:*)       # benchenv:NAME=VALUE:clips — one quick bench line under an environment switch (in-situ A/B of an option)
      A=${STEP#benchenv:}; E=${A%%:*}; N=${A##*:}
      env $E timeout 600 python bench.py --clips-per-step $N --steps 2 --warmup 1 --no-cpu-baseline --no-extra-reading > $O/${TAG}_benchenv.txt 2>&1
      tail -n 1 $O/${TAG}_benchenv.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['readings']; print('$E clips/step $N:', d['value'], 'fps; inv', r['inversion_s_per_clip'], 'smp', r['sampling_s_per_clip'])" ;;
    benchn:*)
      N=${STEP#benchn:}
      timeout 600 python bench.py --clips-per-step $N --steps 2 --warmup 1 --no-cpu-baseline --no-extra-reading > $O/${TAG}_benchn$N.txt 2>&1
      tail -n 1 $O/${TAG}_benchn$N.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['readings']; print('clips/step $N:', d['value'], 'fps; ms_per_step', d['ms_per_step'], 'inv', r['inversion_s_per_clip'], 'smp', r['sampling_s_per_clip'], 'gemm frac', d['roofline']['frac'], 'attainable', d['roofline']['frac_of_attainable'], 'byte-bound share', d['roofline']['byte_bound_time_share'])" ;;
    smoke)
      timeout 400 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.txt 2>&1; show smoke 3 ;;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o r05 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-extra-reading --prof-samples 0 > $O/${TAG}_prof.log 2>&1 )
      DB=$(find $O/${TAG}_prof -name '*.db' | head -n 1)
      [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats.txt 2>&1
      find $O/${TAG}_prof -type f -size +4M -delete 2>/dev/null
      head -n 24 $O/${TAG}_kernel_stats.txt | cut -c1-180; tail -n 3 $O/${TAG}_kernel_stats.txt | cut -c1-200 ;;
    trace:*)                  # trace:N = the same at N clips per step -> TAG_kernel_stats_bN.txt
      N=${STEP#trace:}
      ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_b$N -o r05 -- python $R/bench.py --clips-per-step $N --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-extra-reading --prof-samples 0 > $O/${TAG}_prof_b$N.log 2>&1 )
      DB=$(find $O/${TAG}_prof_b$N -name '*.db' | head -n 1)
      [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_stats_b$N.txt 2>&1
      find $O/${TAG}_prof_b$N -type f -size +4M -delete 2>/dev/null
      head -n 24 $O/${TAG}_kernel_stats_b$N.txt | cut -c1-180; tail -n 3 $O/${TAG}_kernel_stats_b$N.txt | cut -c1-200 ;;
    pmcshape|pmcshape:*)      # pmcshape:N = N clips per step (default 1 = the headline); the entry lands in gpurun_out/TAG_gemm_hbm_traffic_bN.json
      N=1; [ "$STEP" != pmcshape ] && N=${STEP#pmcshape:}
      bash tools/pmc_by_shape.sh ${TAG}_pmc_shape_b$N --clips-per-step $N > $O/${TAG}_pmc_shape_b$N.txt 2>&1; show pmc_shape_b$N 2
      cp $O/${TAG}_pmc_shape_b$N/gemm_hbm_traffic.json $O/${TAG}_gemm_hbm_traffic_b$N.json 2>/dev/null
      cp $O/${TAG}_pmc_shape_b$N/by_shape.txt $O/${TAG}_gemm_traffic_by_shape_b$N.txt 2>/dev/null
      python tools/pmc_by_shape.py --merge $R/profiles/gemm_hbm_traffic.json $O/${TAG}_gemm_hbm_traffic_b$N.json ;;
    pmcsq)
      bash tools/pmc_sq.sh ${TAG}_pmc_sq; cat $O/${TAG}_pmc_sq/passes.txt; head -n 24 $O/${TAG}_pmc_sq/summary.txt | cut -c1-200 ;;
    attn)
      timeout 300 python tools/attn_ab.py > $O/${TAG}_attn.txt 2>&1; show attn 30 ;;
    train)
      timeout 300 python tools/train_bench.py --steps 3 > $O/${TAG}_train.txt 2>&1; show train 4 ;;
    dist1)
      VSX_FORCE_DISTRIBUTED=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline > $O/${TAG}_dist1_clip.txt 2>&1; show dist1_clip 2 3000
      VSX_FORCE_DISTRIBUTED=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 \
        bench.py --gpus 1 --config 4 --frames 32 --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline > $O/${TAG}_dist1_long.txt 2>&1; show dist1_long 2 3000 ;;
    *) echo "unknown step $STEP" ;;
  esac
  el $STEP
done
