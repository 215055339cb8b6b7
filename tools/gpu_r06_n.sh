#!/bin/bash
out=gpurun_out/r06n; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 1500 python -m pytest tests/test_module_gpu.py tests/test_fused_gpu.py tests/test_attn_vfuse_gpu.py -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
ROWS=12 bash tools/prof.sh r06n_one --steps 20 --in-flight 1 > $out/prof.txt
python - <<PY
import csv,statistics
rows=list(csv.DictReader(open('gpurun_out/r06n_one/runc_kernel_trace.csv')))
for k in ('k_attn_kvh','k_attn_o16','k_attn_q16'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if k in r['Kernel_Name']]
    print(k, "odd/even medians %.1f %.1f" % (statistics.median(d[0::2]), statistics.median(d[1::2])))
PY
b one --in-flight 1
b one2 --in-flight 1
b pipe --steps 50
b s20 --steps 20 --warmup 5
