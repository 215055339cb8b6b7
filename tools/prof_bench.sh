#!/bin/bash
# usage: tools/prof_bench.sh <name>   -> rocprofv3 kernel stats of `python3 bench.py` (default arguments, CPU baseline skipped)
name=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- \
    python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/$name.log 2>&1
f=$(ls $root/gpurun_out/$name/*kernel_stats.csv $root/gpurun_out/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
head -8 "$f" | cut -c1-150
