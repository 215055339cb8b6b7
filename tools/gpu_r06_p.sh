#!/bin/bash
out=gpurun_out/r06final; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', d['n_gpus'], d.get('timing',{}).get('median_ms'), (d.get('one_frame_in_flight') or {}).get('value'))"; }
timeout 400 python bench.py --from-points --no-cpu-baseline > $out/bench_from_points_line.json 2>> $out/bench.err; line $out/bench_from_points_line.json
timeout 400 python bench.py --from-points --in-flight 1 --no-cpu-baseline > $out/bench_from_points_one_in_flight_line.json 2>> $out/bench.err; line $out/bench_from_points_one_in_flight_line.json
ROWS=60 bash tools/prof.sh r06final_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; grep "k_ps_\|k_vox\|k_dense" $out/prof_pts_one.txt | awk -F, '{print substr($1,1,24), $4}'
timeout 400 python bench.py --train --detector --sync-bn --batch 2 --steps 5 --warmup 2 > $out/bench_train_detector_b2_line.json 2>> $out/bench.err; line $out/bench_train_detector_b2_line.json
