#!/bin/bash
# MFMA utilisation / stall / LDS-conflict counters of the vsx kernels (north_star: "rocprof ... MFMA utilisation against
# gfx950 peak").  Counter passes are run on their own with --kernel-trace only, as gpurun requires (no sys/runtime/hip
# trace domains next to --pmc).  SQ has 8 slots per pass, GRBM 2; every pass is a separate process so that an unknown
# counter name costs that pass only.  Output: $R/gpurun_out/$TAG/<pass>/*counter_collection.csv; summarise with
#   python tools/pmc_sq_summary.py gpurun_out/$TAG > profiles/rNN_pmc_sq.txt
TAG=${1:-pmc_sq}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-extra-reading --prof-samples 0"
mkdir -p $R/gpurun_out/$TAG
run_pass() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/$TAG/$name -o p --output-format csv -- $CMD \
      > $R/gpurun_out/$TAG/$name.log 2>&1
  echo "pass $name rc=$?" >> $R/gpurun_out/$TAG/passes.txt
}
run_pass mfma  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run_pass stall SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run_pass lds   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_LDS
# per-pass CSVs are large (one row per dispatch and counter): keep the summary inputs only
python $R/tools/pmc_sq_summary.py $R/gpurun_out/$TAG > $R/gpurun_out/$TAG/summary.txt 2>&1
find $R/gpurun_out/$TAG -name '*.csv' -size +8M -delete
