#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VSX_PP_SCHED=8 bash tools/pmc_by_shape.sh r03i_pmc_shape_s8 > gpurun_out/r03i_s8.txt 2>&1
grep -E "geglu|->3840|->1920|total" gpurun_out/r03i_s8.txt | cut -c1-150
