#!/bin/bash
# usage: tools/prof_train.sh <name>  -> rocprofv3 kernel stats of the compact training step
name=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
MSSVT_TRAIN_ONLY_COMPACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- \
    python3 $root/tools/train_time.py > $root/gpurun_out/$name.log 2>&1
f=$(ls $root/gpurun_out/$name/*kernel_stats.csv $root/gpurun_out/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
head -${ROWS:-30} "$f" | cut -c1-200
tail -3 $root/gpurun_out/$name.log
