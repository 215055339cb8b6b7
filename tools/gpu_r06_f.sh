#!/bin/bash
out=gpurun_out/r06f; mkdir -p $out
timeout 600 python tools/pipe_trace.py --steps 20 > $out/pipe_trace_20.txt 2>&1; grep -v amdgpu $out/pipe_trace_20.txt
timeout 600 python tools/pipe_trace.py --steps 50 > $out/pipe_trace_50.txt 2>&1; grep -v amdgpu $out/pipe_trace_50.txt | grep rep
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
MSSVT_XCD_REMAP=1 b one_r1 --in-flight 1
MSSVT_XCD_REMAP=2 b one_r2 --in-flight 1
MSSVT_XCD_REMAP=0 b one_r0 --in-flight 1
MSSVT_XCD_REMAP=1 b one_r1b --in-flight 1
MSSVT_XCD_REMAP=2 b one_r2b --in-flight 1
MSSVT_XCD_REMAP=0 b one_r0b --in-flight 1
MSSVT_PIPE_STREAMS=cumask MSSVT_XCD_REMAP=1 b pipe_r1 --steps 50
MSSVT_PIPE_STREAMS=cumask MSSVT_XCD_REMAP=2 b pipe_r2 --steps 50
MSSVT_PIPE_STREAMS=cumask MSSVT_XCD_REMAP=0 b pipe_r0 --steps 50
MSSVT_XCD_REMAP=2 bash tools/pmc_frame.sh r06f_r2 > $out/pmc_r2.txt 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06f_r2_pmc_frame.json"))
for k in ("k_ffn_ws<128, 256, true, true>","k_attn_kvh<64, 16, 4, 2, true>"):
    e=d.get(k,{})
    print("r2",k,e.get("hbm_bytes_per_launch"),e.get("cycles_per_launch"))
PY
