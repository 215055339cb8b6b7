#!/bin/bash
out=gpurun_out/r06h; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 3000 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest gpu rc $?"; tail -6 $out/pytest_gpu.txt
for t in 16 32 64; do
  MSSVT_PFN_TASK=$t b pts_one_t$t --from-points --in-flight 1
  MSSVT_PFN_TASK=$t b pts_t$t --from-points
done
for t in 16 32; do
MSSVT_PFN_TASK=$t ROWS=60 bash tools/prof.sh r06h_pts_t$t --steps 20 --in-flight 1 --from-points > $out/prof_pts_t$t.txt; grep "k_ps_pfn2\|k_ps_place" $out/prof_pts_t$t.txt | cut -c1-100
done
