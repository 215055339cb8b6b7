#!/bin/bash
# usage: tools/pmc.sh <name> "<COUNTER ...>" [script args]  (run on the GPU box; one --pmc pass, no tracing besides kernel-trace)
name=$1; ctr=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/$name -o runc -- \
    python3 $root/tools/${PMC_PROG:-prof_attn.py} "$@" > $root/gpurun_out/$name.log 2>&1
python3 $root/tools/pmc_summary.py $root/gpurun_out/$name
