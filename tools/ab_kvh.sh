#!/bin/bash
# A/B of compile-time variants of k_attn_kvh on ONE box: tools/ab_kvh.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
    MSSVT_EXTRA_HIPCC_FLAGS="$flags" python -m mssvt_amd.build --force > /dev/null 2>&1
    echo "flags: $flags"
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$RANDOM -o r -- python3 $GRAFT_REPO_ROOT/tools/time_attn.py > /tmp/ab.log 2>&1
    grep "kv16+qo16:" /tmp/ab.log
    cd $GRAFT_REPO_ROOT
    f=$(ls -t /tmp/ab_*/*kernel_stats.csv /tmp/ab_*/*/*kernel_stats.csv 2>/dev/null | head -1)
    grep "k_attn_kvh\|k_attn_q16\|k_attn_o16" "$f" | cut -d, -f1-4 | cut -c1-100
done
