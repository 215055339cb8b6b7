#!/bin/bash
# copy the evidence of tools/gpu_final_r06.sh from the scratch gpurun_out/ into the tracked profiles/ under a tag
# usage: bash tools/collect_profiles_r06.sh r06_z
set -e
tag=${1:?tag}; src=gpurun_out/r06final
for f in $src/*_line.json; do
    n=$(basename $f); grep '^{' $f | tail -1 > profiles/${tag}_$n
done
cp gpurun_out/r06final_b1/runc_kernel_stats.csv profiles/${tag}_b1_bench_command_kernel_stats.csv
cp gpurun_out/r06final_b1_one/runc_kernel_stats.csv profiles/${tag}_b1_one_in_flight_kernel_stats.csv
cp gpurun_out/r06final_pts_one/runc_kernel_stats.csv profiles/${tag}_from_points_one_in_flight_kernel_stats.csv 2>/dev/null || cp $src/prof_pts_one.txt profiles/${tag}_from_points_one_in_flight_kernel_stats.csv
cp gpurun_out/r06final_b8_bf16/runc_kernel_stats.csv profiles/${tag}_b8_bf16_bench_command_kernel_stats.csv
cp gpurun_out/r06final_pmc_frame.json profiles/${tag}_pmc_frame.json
cp gpurun_out/r06final_pmc_frame.json profiles/pmc_frame.json   # the file fused.roofline() reads
cp gpurun_out/r06final_breakdown.json profiles/${tag}_pmc_breakdown.json
cp gpurun_out/r06final_ceiling_pmc.json profiles/${tag}_ceiling_pmc.json
for n in host_frame train_time two_streams pipe_trace ceiling_mix; do grep -v amdgpu.ids $src/$n.txt > profiles/${tag}_$n.txt; done
f=$(ls gpurun_out/r06final_train/*kernel_stats.csv gpurun_out/r06final_train/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/${tag}_train_step_kernel_stats.csv
ls profiles | grep "^${tag}_"
