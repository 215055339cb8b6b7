#!/bin/bash
# round 6, first GPU pass: the new pipeline / schedule tests, the pipeline's call modes, depth by batch size, baseline kernel stats
out=gpurun_out/r06a; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_mfma_schedules_gpu.py -x -q -s > $out/pytest_new.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_new.txt
timeout 600 python tools/two_streams.py --streams 4 > $out/two_streams.txt 2>&1; grep -v amdgpu.ids $out/two_streams.txt
for b in 2 4 8; do for d in 1 2 4; do
  timeout 400 python bench.py --batch $b --in-flight $d --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b${b}_d${d}_line.json 2>> $out/bench.err; line $out/bench_b${b}_d${d}_line.json
done; done
ROWS=45 bash tools/prof.sh r06a_b1_one --steps 20 --in-flight 1 > $out/prof_b1_one.txt; head -40 $out/prof_b1_one.txt | cut -c1-150
ROWS=45 bash tools/prof.sh r06a_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; head -45 $out/prof_pts_one.txt | cut -c1-150
