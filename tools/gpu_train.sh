#!/bin/bash
# training-path check on the GPU box: the training tests, the one-scene step time, kernel stats of the step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_path_gpu.py tests/test_module_gpu.py -q -x -k "train or backward or gradient or segment or csr or linear or pair or layer_norm" 2>&1 | tail -5
MSSVT_TRAIN_ONLY_COMPACT=1 timeout 300 python tools/train_time.py 2>&1 | tail -2 | tee gpurun_out/train_time.txt
if [ -n "$TRAIN_STATS" ]; then
  ROWS=1 bash tools/prof_train.sh train_prof > /dev/null 2>&1
  f=$(ls gpurun_out/train_prof/*kernel_stats.csv gpurun_out/train_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
  cp "$f" gpurun_out/train_step_kernel_stats.csv
  python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/train_step_kernel_stats.csv')))
print("total ms / 7 steps:", sum(int(r['TotalDurationNs']) for r in rows)/1e6, "launches:", sum(int(r['Calls']) for r in rows))
P
fi
