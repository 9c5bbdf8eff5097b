#!/bin/bash
# HBM traffic of the dominant kernel (vsx_gemm_f16) from the L2 fabric counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no tracing domains besides --kernel-trace), FETCH_SIZE doubled on
# gfx950 for wide coalesced reads.  Writes gpurun_out/pmc_traffic/{fetch,write}/..._counter_collection.csv; summarise with
#   python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic > profiles/rNN_gemm_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 1 --warmup 0 --ddim-steps 1 --no-cpu-baseline --no-graphs --prof-samples 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_traffic/fetch -o p --output-format csv -- $CMD > $R/gpurun_out/pmc_traffic_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_traffic/write -o p --output-format csv -- $CMD > $R/gpurun_out/pmc_traffic_write.log 2>&1
python $R/tools/pmc_traffic_summary.py $R/gpurun_out/pmc_traffic > $R/gpurun_out/pmc_traffic/r02_gemm_hbm_traffic.json
find $R/gpurun_out/pmc_traffic -name '*.csv' -size +8M -delete
ls -la $R/gpurun_out/pmc_traffic/*/
