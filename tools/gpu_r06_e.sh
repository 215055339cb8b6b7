#!/bin/bash
out=gpurun_out/r06e; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
timeout 900 python -m pytest tests/test_vfe_gpu.py -x -q > $out/pytest_vfe.txt 2>&1; echo "pytest vfe rc $?"; tail -3 $out/pytest_vfe.txt
ROWS=45 bash tools/prof.sh r06e_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; grep "k_ps_\|k_vox\|k_dense" $out/prof_pts_one.txt | cut -c1-120
b from_points --from-points
for k in priority cumask priority cumask; do
  MSSVT_PIPE_STREAMS=$k b s20_$k --steps 20 --warmup 5
  MSSVT_PIPE_STREAMS=$k b s50_$k --steps 50
done
b one --in-flight 1
MSSVT_XCD_REMAP=0 b one_r0 --in-flight 1
timeout 600 python bench.py --no-cpu-baseline > $out/bench_roofline.json 2>> $out/bench.err; line $out/bench_roofline.json
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06e/bench_roofline.json') if l.startswith('{')][-1])
r=d['roofline']
print({k: r.get(k) for k in ('frac','avg_launch_us','ceiling_us','traffic')}, r.get('ceiling'))
for o in r['other_kernels']:
    print(o.get('cbs_pattern'), o.get('avg_launch_us'), o.get('ceiling_us_window_launch'), o.get('ceiling_us_window_launch_wv_fused'))
PY
tail -3 $out/bench.err
