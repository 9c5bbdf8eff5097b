cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in conv lin; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pmc_a_$k -o p --output-format csv -- python $R/tools/gemm_one.py $k 6 > $R/gpurun_out/pmc_a_$k.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/pmc_b_$k -o p --output-format csv -- python $R/tools/gemm_one.py $k 6 > $R/gpurun_out/pmc_b_$k.log 2>&1
done
ls -R $R/gpurun_out/pmc_a_conv | head
