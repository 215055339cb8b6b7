#!/bin/bash
out=gpurun_out/r06i; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_vfe_gpu.py tests/test_bench_gpu.py -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
b pts --from-points
b pts_s20 --from-points --steps 20 --warmup 5
b pts_d6 --from-points --in-flight 6
b pts_d3 --from-points --in-flight 3
b pts_one --from-points --in-flight 1
b pts2 --from-points
tail -3 $out/bench.err
