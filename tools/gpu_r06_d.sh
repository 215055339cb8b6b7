#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
line() { python -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{')][-1]);print('$1',round(d['value'],1),d['unit'],round(d['ms_per_step'],3),'ms', (d.get('one_frame_in_flight') or {}).get('value'))"; }
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-roofline "$@" > $out/$name.json 2>> $out/bench.err; line $out/$name.json; }
for k in priority cumask; do
  MSSVT_PIPE_STREAMS=$k b s20_$k --steps 20 --warmup 5
  MSSVT_PIPE_STREAMS=$k b s50_$k --steps 50
  MSSVT_PIPE_STREAMS=$k b s20b_$k --steps 20 --warmup 5
done
MSSVT_XCD_REMAP=0 MSSVT_FRAME_SIDE_COPY=0 b one_r0_c0 --in-flight 1
MSSVT_XCD_REMAP=1 MSSVT_FRAME_SIDE_COPY=0 b one_r1_c0 --in-flight 1
MSSVT_XCD_REMAP=0 MSSVT_FRAME_SIDE_COPY=1 b one_r0_c1 --in-flight 1
MSSVT_XCD_REMAP=1 MSSVT_FRAME_SIDE_COPY=1 b one_r1_c1 --in-flight 1
MSSVT_XCD_REMAP=0 b pipe_r0 --steps 50
MSSVT_XCD_REMAP=1 b pipe_r1 --steps 50
MSSVT_XCD_REMAP=0 bash tools/pmc_frame.sh r06d_r0 > $out/pmc_r0.txt 2>&1
MSSVT_XCD_REMAP=1 bash tools/pmc_frame.sh r06d_r1 > $out/pmc_r1.txt 2>&1
python - <<'PY'
import json
for t in ("r0","r1"):
    d=json.load(open("gpurun_out/r06d_%s_pmc_frame.json"%t))
    for k in ("k_ffn_ws<128, 256, true, true>","k_attn_kvh<64, 16, 4, 2, true>","k_attn_o16<64, 4>","k_cmp_ws<128>"):
        e=d.get(k,{})
        print(t,k,e.get("hbm_bytes_per_launch"),e.get("FETCH_SIZE_KB"),e.get("WRITE_SIZE_KB"),e.get("cycles_per_launch"))
PY
timeout 900 python -m pytest tests/test_vfe_gpu.py -x -q > $out/pytest_vfe.txt 2>&1; echo "pytest vfe rc $?"; tail -3 $out/pytest_vfe.txt
ROWS=45 bash tools/prof.sh r06d_pts_one --steps 20 --in-flight 1 --from-points > $out/prof_pts_one.txt; grep "k_ps_\|k_vox\|k_dense" $out/prof_pts_one.txt | cut -c1-120
b from_points --from-points
bash tools/pmc_breakdown.sh r06d > $out/breakdown.txt 2>&1; tail -12 $out/breakdown.txt | cut -c1-1200
