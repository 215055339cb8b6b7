#!/bin/bash
out=gpurun_out/r06final; mkdir -p $out
timeout 3300 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest gpu rc $?"; tail -4 $out/pytest_gpu.txt
bash tools/gpu_final_r06.sh
