#!/bin/bash
out=gpurun_out/r06j; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 900 python -m pytest tests/test_module_gpu.py -q -k "full_size or ffn or golden" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
b one --in-flight 1
b one2 --in-flight 1
b pipe --steps 50
b s20 --steps 20 --warmup 5
ROWS=12 bash tools/prof.sh r06j_b1_one --steps 20 --in-flight 1 > $out/prof_b1_one.txt; head -8 $out/prof_b1_one.txt | cut -c1-150
bash tools/pmc_breakdown.sh r06j > $out/breakdown.txt 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06j_breakdown.json"))
for k in ("k_ffn_ws<128, 256, true, true>","k_ffn_ws<128, 256, false, false>"):
    e=d.get(k,{})
    print(k,{x:e.get(x) for x in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","share_wait_any","share_wait_inst_any")})
PY
