#!/bin/bash
# end-of-round measurements (round 4): bench lines, rocprofv3 kernel stats of the bench commands, PMC counters of the frame
out=gpurun_out/r04final; mkdir -p $out
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc $?"
timeout 400 python bench.py --batch 8 --attn-dtype bf16 --no-cpu-baseline --steps 20 > $out/bench_b8_bf16_line.json 2>> $out/bench.err; echo "b8 bf16 rc $?"
timeout 400 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b8_f32_line.json 2>> $out/bench.err; echo "b8 f32 rc $?"
timeout 400 python bench.py --batch 4 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b4_f32_line.json 2>> $out/bench.err; echo "b4 f32 rc $?"
timeout 400 python bench.py --points 300000 --cfg mssvt_amd/cfgs/mssvt_enlarged.yaml --no-cpu-baseline --no-roofline --steps 20 > $out/bench_enlarged_300k_line.json 2>> $out/bench.err; echo "enlarged rc $?"
timeout 400 python bench.py --train --batch 4 --steps 5 --warmup 2 > $out/bench_train_b4_line.json 2>> $out/bench.err; echo "train rc $?"
for f in $out/*_line.json; do python -c "import json,sys;d=json.loads([l for l in open('$f') if l.startswith('{')][-1]);print('$f',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', d['n_gpus'], d.get('timing',{}).get('median_ms'))"; done
ROWS=45 bash tools/prof.sh r04final_b1 --steps 20 > $out/prof_b1.txt; head -3 $out/prof_b1.txt | cut -c1-150
ROWS=45 bash tools/prof.sh r04final_b8_bf16 --batch 8 --attn-dtype bf16 --steps 10 --warmup 3 > $out/prof_b8.txt; head -3 $out/prof_b8.txt | cut -c1-150
bash tools/pmc_frame.sh r04final > $out/pmc.txt 2>&1; tail -c 600 $out/pmc.txt
bash tools/pmc_frame.sh r04final_b8 8 bf16 > $out/pmc_b8.txt 2>&1; tail -c 300 $out/pmc_b8.txt
timeout 300 python tools/host_frame.py > $out/host_frame.txt 2>&1; tail -9 $out/host_frame.txt
MSSVT_FRAME=0 timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_python_path_line.json 2>> $out/bench.err; echo "python path rc $?"
bash tools/prof_train.sh r04final_train > $out/prof_train.txt 2>&1; head -5 $out/prof_train.txt | cut -c1-150
timeout 300 python tools/train_time.py > $out/train_time.txt 2>&1; tail -2 $out/train_time.txt
