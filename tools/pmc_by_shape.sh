#!/bin/bash
# Per-shape HBM traffic of vsx_gemm_f16 (tools/pmc_by_shape.py): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (kernel trace only, as MI355X_MICROARCH.md prescribes) + one plain kernel trace for undisturbed durations; the product
# logs its GEMM launches in order (VSX_GEMM_LOG) during the first pass.  One inversion step + one CFG step.
#   bash tools/pmc_by_shape.sh [outdir-name] [extra bench.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/${1:-pmc_shape}
shift
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-extra-reading --prof-samples 0 $*"
VSX_GEMM_LOG=$D/gemm_log.txt rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $D/fetch -o p --output-format csv -- $CMD > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $D/write -o p --output-format csv -- $CMD > $D/write.log 2>&1
rocprofv3 --kernel-trace -d $D/trace -o p --output-format csv -- $CMD > $D/trace.log 2>&1
python $R/tools/pmc_by_shape.py $D > $D/by_shape.txt 2>&1
find $D -name '*.csv' -size +6M -delete
head -n 45 $D/by_shape.txt | cut -c1-230
tail -n 1 $D/by_shape.txt
