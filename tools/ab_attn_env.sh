#!/bin/bash
# per-kernel times of one Block attention under different environments on ONE box: tools/ab_attn_env.sh "VAR=1" "VAR=2" ...
for e in "$@"; do
    echo "env: $e"
    d=/tmp/abe_$RANDOM
    cd /tmp && env $e TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $d -o r -- python3 $GRAFT_REPO_ROOT/tools/time_attn.py > /tmp/abe.log 2>&1
    cd $GRAFT_REPO_ROOT
    grep "kv16+qo16:" /tmp/abe.log
    f=$(ls $d/*kernel_trace.csv $d/*/*kernel_trace.csv 2>/dev/null | head -1)
    python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the tool runs modes f32, kv16, kv16+qo16, bf16 for block 0 then block 1: report the 16-form kernels split by position
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'k_attn_q16' in n or 'k_attn_o16' in n or ('k_attn_kvh' in n):
        agg[n.split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    h = len(v) // 2
    print("   %-40s n=%3d  first half (odd pattern) %.1f us   second half (even) %.1f us" % (k[-40:], len(v), sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1)))
PY
done
