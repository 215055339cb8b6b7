#!/bin/bash
# end-of-round measurements (round 6): bench lines, rocprofv3 kernel stats of the bench commands, PMC counters of the frame
out=gpurun_out/r06final; mkdir -p $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', d['n_gpus'], d.get('timing',{}).get('median_ms'), (d.get('one_frame_in_flight') or {}).get('value'))"; }
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc $?"; line $out/bench_line.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_steps20_line.json 2>> $out/bench.err; line $out/bench_driver_steps20_line.json
timeout 400 python bench.py --in-flight 1 --no-cpu-baseline > $out/bench_one_in_flight_line.json 2>> $out/bench.err; line $out/bench_one_in_flight_line.json
timeout 400 python bench.py --arith f32 --no-cpu-baseline > $out/bench_arith_f32_line.json 2>> $out/bench.err; line $out/bench_arith_f32_line.json
timeout 400 python bench.py --arith f32 --in-flight 1 --no-cpu-baseline --no-roofline > $out/bench_arith_f32_one_in_flight_line.json 2>> $out/bench.err; line $out/bench_arith_f32_one_in_flight_line.json
timeout 400 python bench.py --from-points --no-cpu-baseline > $out/bench_from_points_line.json 2>> $out/bench.err; line $out/bench_from_points_line.json
timeout 400 python bench.py --from-points --in-flight 1 --no-cpu-baseline > $out/bench_from_points_one_in_flight_line.json 2>> $out/bench.err; line $out/bench_from_points_one_in_flight_line.json
timeout 400 python bench.py --batch 8 --attn-dtype bf16 --no-cpu-baseline --steps 20 > $out/bench_b8_bf16_line.json 2>> $out/bench.err; line $out/bench_b8_bf16_line.json
timeout 400 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b8_f32_line.json 2>> $out/bench.err; line $out/bench_b8_f32_line.json
timeout 400 python bench.py --batch 4 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b4_f32_line.json 2>> $out/bench.err; line $out/bench_b4_f32_line.json
timeout 400 python bench.py --points 300000 --cfg mssvt_amd/cfgs/mssvt_enlarged.yaml --no-cpu-baseline --no-roofline --steps 20 > $out/bench_enlarged_300k_line.json 2>> $out/bench.err; line $out/bench_enlarged_300k_line.json
timeout 400 python bench.py --train --batch 4 --steps 5 --warmup 2 > $out/bench_train_b4_line.json 2>> $out/bench.err; line $out/bench_train_b4_line.json
timeout 400 python bench.py --train --detector --sync-bn --batch 2 --steps 5 --warmup 2 > $out/bench_train_detector_b2_line.json 2>> $out/bench.err; line $out/bench_train_detector_b2_line.json
ROWS=45 bash tools/prof.sh r06final_b1 --steps 20 > $out/prof_b1.txt; head -3 $out/prof_b1.txt | cut -c1-150
ROWS=45 bash tools/prof.sh r06final_b1_one --steps 20 --in-flight 1 > $out/prof_b1_one.txt; head -3 $out/prof_b1_one.txt | cut -c1-150
ROWS=60 bash tools/prof.sh r06final_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; head -3 $out/prof_pts_one.txt | cut -c1-150
ROWS=45 bash tools/prof.sh r06final_b8_bf16 --batch 8 --attn-dtype bf16 --steps 10 --warmup 3 > $out/prof_b8.txt; head -3 $out/prof_b8.txt | cut -c1-150
bash tools/pmc_frame.sh r06final > $out/pmc.txt 2>&1; tail -c 600 $out/pmc.txt
bash tools/pmc_breakdown.sh r06final > $out/breakdown.txt 2>&1; tail -c 400 $out/breakdown.txt
bash tools/pmc_ceiling.sh r06final > $out/pmc_ceiling.txt 2>&1; tail -6 $out/pmc_ceiling.txt | cut -c1-300
timeout 600 python tools/ceiling_mix.py > $out/ceiling_mix.txt 2>&1; grep -v amdgpu $out/ceiling_mix.txt | head -3
timeout 300 python tools/host_frame.py > $out/host_frame.txt 2>&1; tail -9 $out/host_frame.txt
bash tools/prof_train.sh r06final_train > $out/prof_train.txt 2>&1; head -5 $out/prof_train.txt | cut -c1-150
timeout 300 python tools/train_time.py > $out/train_time.txt 2>&1; tail -2 $out/train_time.txt
timeout 600 python tools/two_streams.py --streams 4 --kinds cumask,priority,pooled > $out/two_streams.txt 2>&1; grep Pipeline $out/two_streams.txt
timeout 300 python tools/pipe_trace.py --steps 20 > $out/pipe_trace.txt 2>&1; grep rep $out/pipe_trace.txt
