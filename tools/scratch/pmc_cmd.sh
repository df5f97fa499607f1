cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STROTSS_WINOGRAD_TILE=4 STROTSS_WINO_FUSED=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcA -o f -- python3 tools/conv_bench.py 1024 3 > gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmcB -o f -- python3 tools/conv_bench.py 1024 3 > gpurun_out/pmcB.log 2>&1 &&
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcC -o f -- python3 tools/conv_bench.py 1024 3 > gpurun_out/pmcC.log 2>&1
echo rc=$?
