#!/bin/bash
out=gpurun_out/r06q; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 900 python -m pytest tests/test_vfe_gpu.py -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -2 $out/pytest.txt
ROWS=60 bash tools/prof.sh r06q_pts --steps 20 --in-flight 1 --from-points > $out/prof.txt; grep "k_ps_pfn2\|k_dense" $out/prof.txt | awk -F, '{print substr($1,1,24), $4}'
b pts --from-points
b pts_one --from-points --in-flight 1
for ch in 16 64; do
  cp mssvt_amd/lib/libmssvt_hip.so /tmp/lib_keep.so
  touch mssvt_amd/csrc/dense_bev.hip
  MSSVT_EXTRA_HIPCC_FLAGS="-DDB4_CH=$ch" python -m mssvt_amd.build > $out/build_$ch.txt 2>&1
  ROWS=60 bash tools/prof.sh r06q_ch$ch --steps 20 --in-flight 1 --from-points > $out/prof_ch$ch.txt; echo "DB4_CH=$ch"; grep "k_dense" $out/prof_ch$ch.txt | awk -F, '{print substr($1,1,24), $4}'
  cp /tmp/lib_keep.so mssvt_amd/lib/libmssvt_hip.so
done
