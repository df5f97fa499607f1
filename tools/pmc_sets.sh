#!/bin/bash
# SQ counter sets of the step's kernels on the GPU box: tools/pmc_sets.sh rNN  (writes gpurun_out/pmc_rNN/summary.txt -> copy to profiles/)
# two separate --pmc passes of `bench.py --no-families --no-graph --no-long-window --steps 6 --warmup 2` (counters serialise the kernels)
r=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$r
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-pyramid --no-families --no-graph --no-live-pmc --steps 6"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/A -- python3 $B > /dev/null 2> $out/A.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d $out/C -- python3 $B > /dev/null 2> $out/C.err || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/A $out/C > $out/summary.txt
find $out -name "*counter_collection.csv" -size +8M -delete; find $out -name "*_kernel_trace.csv" -size +8M -delete
tail -40 $out/summary.txt
