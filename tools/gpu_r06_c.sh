#!/bin/bash
out=gpurun_out/r06c; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'), (d.get('from_points') or {}).get('frac'))"; }
timeout 1500 python -m pytest tests/test_vfe_gpu.py tests/test_pipeline_gpu.py tests/test_detector_gpu.py -x -q -s > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.txt
timeout 600 python -m pytest tests/ -x -q -m gpu -k "dense or height" > $out/pytest_dense.txt 2>&1; echo "pytest dense rc $?"; tail -3 $out/pytest_dense.txt
timeout 400 python bench.py --from-points --no-cpu-baseline > $out/bench_from_points_line.json 2>> $out/bench.err; line $out/bench_from_points_line.json
timeout 400 python bench.py --from-points --in-flight 1 --no-cpu-baseline > $out/bench_from_points_one_line.json 2>> $out/bench.err; line $out/bench_from_points_one_line.json
ROWS=45 bash tools/prof.sh r06c_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; head -36 $out/prof_pts_one.txt | cut -c1-150
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_line.json 2>> $out/bench.err; line $out/bench_driver_line.json
tail -5 $out/bench.err
