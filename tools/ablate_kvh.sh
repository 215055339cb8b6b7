#!/bin/bash
# timing-only ablations of k_attn_kvh (results are wrong by construction): which resource bounds the window launch?
# 1 gathers from 8 rows (L1 hits)  2 no Xbar store  3 no Qt loads  4 no second product  5 no score product  6 no token build
for k in 0 1 2 3 4 5 6; do
    if [ $k = 0 ]; then flags=""; else flags="-DKVH_ABLATE=$k"; fi
    MSSVT_EXTRA_HIPCC_FLAGS="$flags" python -m mssvt_amd.build --force > /dev/null 2>&1
    echo "ablation $k"
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$k -o r -- python3 $GRAFT_REPO_ROOT/tools/time_attn.py > /dev/null 2>&1
    cd $GRAFT_REPO_ROOT
    f=$(ls /tmp/abl$k/*kernel_stats.csv /tmp/abl$k/*/*kernel_stats.csv 2>/dev/null | head -1)
    grep "k_attn_kvh" "$f" | cut -d, -f1-4,6,7 | cut -c1-100
done
