cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in conv conv2 lin geglu; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc2_$k -o p --output-format csv -- python $R/tools/gemm_one.py $k 6 > $R/gpurun_out/pmc2_$k.log 2>&1
done
