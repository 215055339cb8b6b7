#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
timeout 600 python -m pytest tests/test_frame_gpu.py -x -q > $out/pytest_frame.txt 2>&1; rc=$?; echo "pytest frame rc $rc"; tail -3 $out/pytest_frame.txt
if [ $rc -ne 0 ]; then grep -B5 -A25 "Error\|assert" $out/pytest_frame.txt | head -60; fi
ROWS=40 bash tools/prof.sh r04g_b1 --steps 20 > $out/prof_b1.txt 2>&1; cut -d, -f1-4 $out/prof_b1.txt | cut -c1-110 | head -28
line() { python -c "
import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),round(d['ms_per_step'],4),'median',d.get('timing',{}).get('median_ms'))"; }
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_$i.json 2>> $out/bench.err; line $out/bench_$i.json; done
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b8.json 2>> $out/bench.err; line $out/bench_b8.json
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
