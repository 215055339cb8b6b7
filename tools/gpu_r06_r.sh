#!/bin/bash
out=gpurun_out/r06r; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 900 python -m pytest tests/test_vfe_gpu.py tests/test_fused_gpu.py -q -k "vfe or dense or sorted or points" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.txt
ROWS=60 bash tools/prof.sh r06r_pts --steps 20 --in-flight 1 --from-points > $out/prof.txt; grep "k_ps_max1\|k_dense\|k_ps_pfn2" gpurun_out/r06r_pts/runc_kernel_stats.csv | awk -F'",' '{print substr($1,1,22), $2}' | cut -c1-70
b pts --from-points
b pts_one --from-points --in-flight 1
