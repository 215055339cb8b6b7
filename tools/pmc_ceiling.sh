#!/bin/bash
# usage: tools/pmc_ceiling.sh <name>   (run on the GPU box) -> instruction counts of the ceiling kernels beside their product kernels
name=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv \
    -d $root/gpurun_out/${name}_ceil -o runc -- python3 $root/tools/prof_ceiling.py > $root/gpurun_out/${name}_ceil.log 2>&1
python3 $root/tools/pmc_frame_summary.py $root/gpurun_out/${name}_ceiling_pmc.json $root/gpurun_out/${name}_ceil | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in sorted(d):
    if any(s in k for s in ('ceiling','k_ffn_ws<128, 256, true','k_attn_kvh')):
        print(k, {x: d[k].get(x) for x in ('launches_seen','SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_SALU','cycles_per_launch','valu_issue_frac','wave_wait_frac')})
"
