#!/bin/bash
out=gpurun_out/r06v; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 1500 python -m pytest tests/test_module_gpu.py tests/test_pipeline_gpu.py tests/test_fused_gpu.py -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
ROWS=12 bash tools/prof.sh r06v_one --steps 20 --in-flight 1 > $out/prof.txt; grep "k_ffn_ws" gpurun_out/r06v_one/runc_kernel_stats.csv | awk -F'",' '{print substr($1,1,48), $2}' | cut -c1-110
b one --in-flight 1
b one2 --in-flight 1
b pipe --steps 50
