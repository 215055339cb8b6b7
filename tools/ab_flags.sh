#!/bin/bash
# A/B of compile-time variants on ONE box: tools/ab_flags.sh <kernel name regex> "<flags A>" "<flags B>" ...
# each variant: full rebuild with MSSVT_EXTRA_HIPCC_FLAGS, rocprofv3 kernel stats of the bench command, the matching rows + the bench line
pat=$1; shift
i=0
for flags in "$@"; do
    i=$((i+1))
    MSSVT_EXTRA_HIPCC_FLAGS="$flags" python -m mssvt_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
    ROWS=60 bash tools/prof.sh ab_v$i --steps ${STEPS:-20} ${BENCH_ARGS} > gpurun_out/ab_v$i.txt 2>&1
    echo "== v$i: $flags"
    grep -E "$pat" gpurun_out/ab_v$i.txt | cut -d, -f1-4 | cut -c1-110
    python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab_v$i.log') if l.startswith('{')][-1]); print("   v$i", round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms')
except Exception as e:
    print("   v$i no bench line", e)
PY
done
python -m mssvt_amd.build --force > /dev/null 2>&1
