#!/bin/bash
# round 4, call A: parity of the new k_ffn_ws loop, A/B of its variants, span stamps, host time
out=gpurun_out/r04a; mkdir -p $out
nproc > $out/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
V0="-DFFW_MANUAL=0 -DFFW_MASKS=0 -DFFW_LN_LDS=0"
V1="-DFFW_MANUAL=0 -DFFW_MASKS=1 -DFFW_LN_LDS=0"
V2="-DFFW_MANUAL=1 -DFFW_MASKS=1 -DFFW_LN_LDS=1"
V3="-DFFW_MANUAL=1 -DFFW_MASKS=1 -DFFW_LN_LDS=0"
V4="-DFFW_MANUAL=0 -DFFW_MASKS=0 -DFFW_LN_LDS=1"
bash tools/ab_flags.sh "k_ffn_ws" "$V2" "$V0" "$V3" "$V1" "$V4" "$V2" "$V0" > $out/ab_ffn.txt 2>&1; cat $out/ab_ffn.txt
for v in "$V2" "$V0"; do
    MSSVT_EXTRA_HIPCC_FLAGS="$v -DMSSVT_STAMPS" python -m mssvt_amd.build --force > /dev/null 2>&1
    echo "== stamps: $v"; timeout 300 python tools/stamps_ffn_ws.py 2>&1 | tail -24
done > $out/stamps.txt 2>&1; cat $out/stamps.txt
python -m mssvt_amd.build --force > /dev/null 2>&1
timeout 300 python tools/host_sections.py > $out/host.txt 2>&1; head -12 $out/host.txt
timeout 600 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc $?"; python -c "
import json;d=json.loads([l for l in open('$out/bench_line.json') if l.startswith('{')][-1]);print(round(d['value'],1),d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'])"
