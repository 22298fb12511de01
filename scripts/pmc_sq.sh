cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r2_counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r2_pmc_a -o p -- python3 scripts/microbench.py igemm > gpurun_out/r2_pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r2_pmc_b -o p -- python3 scripts/microbench.py igemm > gpurun_out/r2_pmc_b.log 2>&1
tail -3 gpurun_out/r2_pmc_a.log gpurun_out/r2_pmc_b.log
