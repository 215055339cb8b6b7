#!/bin/bash
# one GPU session: the -m gpu suite, the default bench line, configs[2]
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_attn_bf16_gpu.py -m gpu -q -s 2>&1 | grep -E "^bf16|attention rows|passed|failed" > gpurun_out/r02/bf16_errors.log
cat gpurun_out/r02/bf16_errors.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02/pytest.log
tail -5 gpurun_out/r02/pytest.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo "bench rc $?"
python -c "import json;d=json.loads(open('gpurun_out/r02/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['timing'])"
for dt in f32 bf16; do
timeout 400 python bench.py --batch 8 --attn-dtype $dt --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r02/bench_b8_$dt.json 2> gpurun_out/r02/bench_b8_$dt.err; echo "bench b8 $dt rc $?"
python -c "import json;d=json.loads(open('gpurun_out/r02/bench_b8_$dt.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['timing'],d['config']['workload'])"
done
