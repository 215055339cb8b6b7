#!/bin/bash
# one GPU session: the -m gpu suite, the default bench line, the multi-rank launcher on one device
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02/pytest.log
tail -5 gpurun_out/r02/pytest.log
timeout 400 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/r02/bench.json
