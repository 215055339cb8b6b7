#!/bin/bash
out=gpurun_out/r06b; mkdir -p $out
timeout 900 python tools/two_streams.py --streams 4 > $out/two_streams.txt 2>&1; grep -v amdgpu.ids $out/two_streams.txt | grep -v "network copy"
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/two_streams.py --streams 4 --kinds priority,pooled > $out/two_streams_hwq8.txt 2>&1; grep Pipeline $out/two_streams_hwq8.txt
timeout 600 python tools/ceiling_mix.py > $out/ceiling_mix.txt 2>&1; grep -v amdgpu.ids $out/ceiling_mix.txt
bash tools/pmc_ceiling_mix.sh r06b > $out/pmc_mix.txt 2>&1; tail -25 $out/pmc_mix.txt
