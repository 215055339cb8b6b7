#!/bin/bash
# round 4, call B: the whole-frame C call -- parity with the Python-driven path, host time, bench with / without it
out=gpurun_out/r04b; mkdir -p $out
timeout 900 python -m pytest tests/test_frame_gpu.py -x -q > $out/pytest_frame.txt 2>&1; echo "pytest frame rc $?"; tail -15 $out/pytest_frame.txt
timeout 300 python tools/host_frame.py > $out/host_frame.txt 2>&1; cat $out/host_frame.txt | tail -10
line() { python -c "
import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),round(d['ms_per_step'],4),'median',d.get('timing',{}).get('median_ms'))"; }
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_frame_$i.json 2>> $out/bench.err; line $out/bench_frame_$i.json
MSSVT_FRAME=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_python_$i.json 2>> $out/bench.err; line $out/bench_python_$i.json
MSSVT_FRAME_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_overlap_$i.json 2>> $out/bench.err; line $out/bench_overlap_$i.json
done
for b in 4 8; do
timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b${b}_frame.json 2>> $out/bench.err; line $out/bench_b${b}_frame.json
MSSVT_FRAME_OVERLAP=1 timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b${b}_overlap.json 2>> $out/bench.err; line $out/bench_b${b}_overlap.json
MSSVT_FRAME=0 timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b${b}_python.json 2>> $out/bench.err; line $out/bench_b${b}_python.json
done
ROWS=50 bash tools/prof.sh r04b_b1 --steps 20 > $out/prof_b1.txt; cat $out/prof_b1.txt | cut -d, -f1-4 | cut -c1-120
MSSVT_FRAME_OVERLAP=1 ROWS=50 bash tools/prof.sh r04b_b1_overlap --steps 20 > $out/prof_b1_overlap.txt; head -4 $out/prof_b1_overlap.txt | cut -c1-100
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
bash tools/ab_flags.sh "k_attn_kvh|k_attn_q16|k_attn_o16" "-DKVH_LAZY=1" "-DKVH_LAZY=0" "-DKVH_LAZY=1" "-DKVH_LAZY=0" > $out/ab_kvh.txt 2>&1; cat $out/ab_kvh.txt
