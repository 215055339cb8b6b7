#!/bin/bash
# frame time and the small kernels' times under different environments on ONE box: tools/ab_env_frame.sh "VAR=1" "VAR=2" ...
for e in "$@"; do
    echo "env: $e"
    env $e python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms, median', round(d['timing']['median_ms'],4))"
    d=/tmp/abf_$RANDOM
    cd /tmp && env $e TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 20 > /dev/null 2>&1
    cd $GRAFT_REPO_ROOT
    f=$(ls $d/*kernel_stats.csv $d/*/*kernel_stats.csv 2>/dev/null | head -1)
    grep "k_plan_order\|k_query_rows" "$f" | cut -d, -f1-4 | cut -c1-90
done
