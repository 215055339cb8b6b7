#!/bin/bash
# A/B of two versions of csrc/ffn.hip on ONE box: tools/ab_ffn.sh <old ffn.hip copy>   (the tree holds the new one)
cp mssvt_amd/csrc/ffn.hip /tmp/ffn_new.hip
for v in new old new old; do
    if [ $v = old ]; then cp "$1" mssvt_amd/csrc/ffn.hip; else cp /tmp/ffn_new.hip mssvt_amd/csrc/ffn.hip; fi
    python -m mssvt_amd.build --force > /dev/null 2>&1
    ROWS=3 bash tools/prof.sh ab_ffn_$v --steps 20 2>&1 | grep "k_ffn_wsILi128ELi256ELb1ELb1" | cut -d, -f1-4 | cut -c1-90
    python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/ab_ffn_$v.log') if l.startswith('{')][-1]); print("$v", round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms')
PY
done
cp /tmp/ffn_new.hip mssvt_amd/csrc/ffn.hip
