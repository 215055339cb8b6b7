#!/bin/bash
# usage: tools/prof_bench.sh <name> [bench args]  -> rocprofv3 kernel stats of the bench command (no cpu baseline)
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o runc -- \
    python3 $root/bench.py --no-cpu-baseline --no-roofline --steps 20 "$@" > $root/gpurun_out/$name.log 2>&1
f=$(ls $root/gpurun_out/$name/*kernel_stats.csv $root/gpurun_out/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
head -${ROWS:-16} "$f" | cut -c1-160
tail -1 $root/gpurun_out/$name.log | cut -c1-300
