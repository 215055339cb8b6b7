#!/bin/bash
# usage: tools/pmc_breakdown.sh <name>   (GPU box): where do the dominant kernels' wave cycles go?  Three --pmc passes over
# whole forwards of the bench frame (tools/prof_frame.py) -> gpurun_out/<name>_breakdown.json (per kernel: mean of every counter)
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/${name}_b$i -o runc -- \
        python3 $root/tools/prof_frame.py "$@" > $root/gpurun_out/${name}_b$i.log 2>&1
done
python3 $root/tools/pmc_breakdown_summary.py $root/gpurun_out/${name}_breakdown.json $root/gpurun_out/${name}_b1 $root/gpurun_out/${name}_b2 $root/gpurun_out/${name}_b3
