#!/bin/bash
# copy the evidence of tools/gpu_final_r04.sh from the scratch gpurun_out/ into the tracked profiles/ under a tag
# usage: bash tools/collect_profiles.sh r03_a
set -e
tag=${1:?tag}; src=gpurun_out/r04final
for f in $src/*_line.json; do
    n=$(basename $f); grep '^{' $f | tail -1 > profiles/${tag}_$n
done
cp gpurun_out/r04final_b1/runc_kernel_stats.csv profiles/${tag}_b1_bench_command_kernel_stats.csv
cp gpurun_out/r04final_b8_bf16/runc_kernel_stats.csv profiles/${tag}_b8_bf16_bench_command_kernel_stats.csv
cp gpurun_out/r04final_pmc_frame.json profiles/${tag}_pmc_frame.json
cp gpurun_out/r04final_pmc_frame.json profiles/pmc_frame.json   # the file fused.roofline() reads
cp gpurun_out/r04final_b8_pmc_frame.json profiles/${tag}_b8_bf16_pmc_frame.json
ls profiles | grep "^${tag}_"
cp gpurun_out/r04final/host_frame.txt profiles/${tag}_host_frame.txt
cp gpurun_out/r04final/train_time.txt profiles/${tag}_train_time.txt
f=$(ls gpurun_out/r04final_train/*kernel_stats.csv gpurun_out/r04final_train/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/${tag}_train_step_kernel_stats.csv
