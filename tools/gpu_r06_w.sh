#!/bin/bash
out=gpurun_out/r06final; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', d['n_gpus'], d.get('timing',{}).get('median_ms'), (d.get('one_frame_in_flight') or {}).get('value'))"; }
timeout 3300 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest gpu rc $?"; tail -3 $out/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc $?"; line $out/bench_line.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_steps20_line.json 2>> $out/bench.err; line $out/bench_driver_steps20_line.json
timeout 400 python bench.py --in-flight 1 --no-cpu-baseline > $out/bench_one_in_flight_line.json 2>> $out/bench.err; line $out/bench_one_in_flight_line.json
timeout 400 python bench.py --batch 8 --attn-dtype bf16 --no-cpu-baseline --steps 20 > $out/bench_b8_bf16_line.json 2>> $out/bench.err; line $out/bench_b8_bf16_line.json
timeout 400 python bench.py --batch 4 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b4_f32_line.json 2>> $out/bench.err; line $out/bench_b4_f32_line.json
ROWS=45 bash tools/prof.sh r06final_b1 --steps 20 > $out/prof_b1.txt; head -3 $out/prof_b1.txt | cut -c1-150
ROWS=45 bash tools/prof.sh r06final_b1_one --steps 20 --in-flight 1 > $out/prof_b1_one.txt; head -3 $out/prof_b1_one.txt | cut -c1-150
bash tools/pmc_frame.sh r06final > $out/pmc.txt 2>&1; tail -c 300 $out/pmc.txt
