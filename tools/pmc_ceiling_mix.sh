#!/bin/bash
# usage: tools/pmc_ceiling_mix.sh <name>   (GPU box) -> instruction counts + durations of every k_ceiling_ffn_mix variant
name=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv \
    -d $root/gpurun_out/${name}_mix -o runc -- python3 $root/tools/ceiling_mix.py --reps 3 > $root/gpurun_out/${name}_mix.log 2>&1
python3 $root/tools/pmc_frame_summary.py $root/gpurun_out/${name}_ceiling_mix_pmc.json $root/gpurun_out/${name}_mix | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in sorted(d):
    if 'ceiling' in k or 'k_ffn_ws<128, 256, true' in k:
        print(k, {x: d[k].get(x) for x in ('launches_seen','SQ_INSTS_VALU','SQ_INSTS_MFMA','cycles_per_launch','valu_issue_frac','wave_wait_frac')})
"
