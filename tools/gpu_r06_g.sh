#!/bin/bash
out=gpurun_out/r06g; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest gpu rc $?"; tail -4 $out/pytest_gpu.txt
MSSVT_XCD_REMAP=1 b one_r1 --in-flight 1
MSSVT_XCD_REMAP=0 b one_r0 --in-flight 1
MSSVT_XCD_REMAP=1 b one_r1b --in-flight 1
MSSVT_XCD_REMAP=0 b one_r0b --in-flight 1
MSSVT_XCD_REMAP=1 b pipe_r1 --steps 50
MSSVT_XCD_REMAP=0 b pipe_r0 --steps 50
MSSVT_XCD_REMAP=1 b pipe_r1b --steps 50
MSSVT_XCD_REMAP=0 b pipe_r0b --steps 50
b s20 --steps 20 --warmup 5
b from_points --from-points
b from_points_one --from-points --in-flight 1
MSSVT_XCD_REMAP=1 bash tools/pmc_frame.sh r06g_r1 > $out/pmc_r1.txt 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06g_r1_pmc_frame.json"))
for k in ("k_ffn_ws<128, 256, true, true>","k_attn_kvh<64, 16, 4, 2, true>","k_ffn_ws<128, 256, false, false>"):
    e=d.get(k,{})
    print("r1",k,e.get("hbm_bytes_per_launch"),e.get("cycles_per_launch"))
PY
