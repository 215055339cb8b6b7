#!/bin/bash
out=gpurun_out/r06m; mkdir -p $out
for v in 0 1 2 3; do
MSSVT_ATTN_VFUSE=$v ROWS=12 bash tools/prof.sh r06m_v$v --steps 20 --in-flight 1 > $out/prof_v$v.txt
python - <<PY
import csv,statistics
rows=list(csv.DictReader(open('gpurun_out/r06m_v$v/runc_kernel_trace.csv')))
for k in ('k_attn_kvh','k_attn_o16'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if k in r['Kernel_Name']]
    print("vfuse=$v", k, "odd/even medians %.1f %.1f" % (statistics.median(d[0::2]), statistics.median(d[1::2])))
PY
done
