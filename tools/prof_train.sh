#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_train -o runc -- python3 $root/tools/train_time.py > $root/gpurun_out/prof_train.log 2>&1
head -24 $root/gpurun_out/prof_train/runc_kernel_stats.csv | cut -c1-150
