#!/bin/bash
# end-of-round measurements: bench lines (configs[1], configs[2], training step) and rocprofv3 kernel stats of the bench commands
mkdir -p gpurun_out/r02g
timeout 600 python bench.py > gpurun_out/r02g/bench_line.json 2> gpurun_out/r02g/bench.err; echo "bench rc $?"
timeout 400 python bench.py --batch 8 --attn-dtype bf16 --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r02g/bench_b8_bf16_line.json 2>> gpurun_out/r02g/bench.err; echo "b8 bf16 rc $?"
timeout 400 python bench.py --batch 8 --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r02g/bench_b8_f32_line.json 2>> gpurun_out/r02g/bench.err; echo "b8 f32 rc $?"
timeout 400 python bench.py --batch 4 --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r02g/bench_b4_f32_line.json 2>> gpurun_out/r02g/bench.err; echo "b4 f32 rc $?"
timeout 400 python bench.py --train --batch 4 --steps 5 --warmup 2 > gpurun_out/r02g/bench_train_b4_line.json 2>> gpurun_out/r02g/bench.err; echo "train rc $?"
MSSVT_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus 2 --train --batch 2 --steps 5 --warmup 2 > gpurun_out/r02g/bench_train_2ranks_one_device_line.json 2>> gpurun_out/r02g/bench.err; echo "train 2 ranks rc $?"
for f in gpurun_out/r02g/*_line.json; do python -c "import json,sys;d=json.loads([l for l in open('$f') if l.startswith('{')][-1]);print('$f',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', d['n_gpus'], d.get('timing',{}).get('median_ms'))"; done
ROWS=40 bash tools/prof.sh r02g_b1 --steps 20 > gpurun_out/r02g/prof_b1.txt; head -3 gpurun_out/r02g/prof_b1.txt | cut -c1-150
ROWS=40 bash tools/prof.sh r02g_b8_bf16 --batch 8 --attn-dtype bf16 --steps 10 --warmup 3 > gpurun_out/r02g/prof_b8.txt; head -3 gpurun_out/r02g/prof_b8.txt | cut -c1-150
