#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
timeout 300 python -m pytest tests/test_fused_gpu.py tests/test_frame_gpu.py -x -q -k "ffn or frame" > $out/pytest_ffn.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $out/pytest_ffn.txt
ROWS=40 bash tools/prof.sh r04f_b1 --steps 20 > $out/prof_b1.txt 2>&1; grep "k_ffn_ws\|kvh\|window_plan<" $out/prof_b1.txt | cut -d, -f1-4 | cut -c1-100
line() { python -c "
import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),round(d['ms_per_step'],4),'median',d.get('timing',{}).get('median_ms'))"; }
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_$i.json 2>> $out/bench.err; line $out/bench_$i.json; done
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b8.json 2>> $out/bench.err; line $out/bench_b8.json
