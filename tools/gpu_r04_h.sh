#!/bin/bash
# the one-launch CompressBlock attention (csrc/compress_ws.hip): its tests, the frame tests, a bench line, kernel stats
out=gpurun_out/r04h; mkdir -p $out
timeout 900 python -m pytest tests/test_compress_ws_gpu.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_frame_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --steps 40 > $out/bench_1.json 2> $out/bench.err; echo "bench rc $?"
MSSVT_CMP_WS=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_ws0.json 2>> $out/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_2.json 2>> $out/bench.err
for f in $out/bench_*.json; do python -c "import json,sys;d=json.loads([l for l in open('$f') if l.startswith('{')][-1]);print('$f',round(d['value'],1),round(d['ms_per_step'],4),'median',d.get('timing',{}).get('median_ms'))"; done
ROWS=30 bash tools/prof.sh r04h_b1 --steps 20 > $out/prof_b1.txt; head -34 $out/prof_b1.txt | cut -c1-140
