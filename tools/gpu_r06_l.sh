#!/bin/bash
out=gpurun_out/r06l; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
MSSVT_ATTN_VFUSE=1 timeout 1500 python -m pytest tests/test_module_gpu.py tests/test_fused_gpu.py -x -q > $out/pytest_vfuse.txt 2>&1; echo "pytest vfuse rc $?"; tail -5 $out/pytest_vfuse.txt
MSSVT_ATTN_VFUSE=0 b one_v0 --in-flight 1
MSSVT_ATTN_VFUSE=1 b one_v1 --in-flight 1
MSSVT_ATTN_VFUSE=0 b one_v0b --in-flight 1
MSSVT_ATTN_VFUSE=1 b one_v1b --in-flight 1
MSSVT_ATTN_VFUSE=0 b pipe_v0 --steps 50
MSSVT_ATTN_VFUSE=1 b pipe_v1 --steps 50
MSSVT_ATTN_VFUSE=1 ROWS=12 bash tools/prof.sh r06l_v1 --steps 20 --in-flight 1 > $out/prof_v1.txt; head -9 $out/prof_v1.txt | cut -c1-130
tail -3 $out/bench.err
