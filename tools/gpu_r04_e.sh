#!/bin/bash
# round 4, call E: k_ffn_ws with buffer-resource addressing; overlap auto; full suite
out=gpurun_out/r04e; mkdir -p $out
timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -k "ffn" > $out/pytest_ffn.txt 2>&1; rc=$?; echo "pytest ffn rc $rc"; tail -3 $out/pytest_ffn.txt
if [ $rc -ne 0 ]; then grep -B5 -A25 "Error\|assert" $out/pytest_ffn.txt | head -80; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
ROWS=40 bash tools/prof.sh r04e_b1 --steps 20 > $out/prof_b1.txt 2>&1; grep "k_ffn_ws\|kvh\|window_plan<" $out/prof_b1.txt | cut -d, -f1-4 | cut -c1-100
line() { python -c "
import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),round(d['ms_per_step'],4),'median',d.get('timing',{}).get('median_ms'))"; }
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_$i.json 2>> $out/bench.err; line $out/bench_$i.json; done
for b in 4 8; do timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b$b.json 2>> $out/bench.err; line $out/bench_b$b.json; done
timeout 300 python bench.py --batch 8 --attn-dtype bf16 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b8_bf16.json 2>> $out/bench.err; line $out/bench_b8_bf16.json
timeout 300 python tools/train_time.py > $out/train.txt 2>&1; tail -2 $out/train.txt
