#!/bin/bash
# usage: tools/prof.sh <name> [bench args]   -> gpurun_out/<name>/runc/*_kernel_stats.csv (run on the GPU box)
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- \
    python3 $root/bench.py --no-cpu-baseline --no-roofline "$@" > $root/gpurun_out/$name.log 2>&1
f=$(ls $root/gpurun_out/$name/*kernel_stats.csv $root/gpurun_out/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
head -${ROWS:-24} "$f" | cut -c1-170
