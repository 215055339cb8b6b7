#!/bin/bash
# usage: tools/pmc_frame.sh <name> [batch] [attn dtype]   (run on the GPU box)
# Three rocprofv3 --pmc passes over whole forwards of the bench frame (tools/prof_frame.py) -- FETCH_SIZE and WRITE_SIZE
# cannot share a pass -- then tools/pmc_frame_summary.py -> gpurun_out/<name>_pmc_frame.json (copy it to profiles/).
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"; do
    i=$((i + 1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/${name}_p$i -o runc -- \
        python3 $root/tools/prof_frame.py "$@" > $root/gpurun_out/${name}_p$i.log 2>&1
done
python3 $root/tools/pmc_frame_summary.py $root/gpurun_out/${name}_pmc_frame.json $root/gpurun_out/${name}_p1 $root/gpurun_out/${name}_p2 $root/gpurun_out/${name}_p3
