#!/bin/bash
# round 4, call C: k_ffn_wsp (producer / consumer waves) -- parity with k_ffn_ws, timing of both forms
out=gpurun_out/r04c; mkdir -p $out
timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -k "ffn" > $out/pytest_ffn.txt 2>&1; rc=$?; echo "pytest ffn rc $rc"; tail -5 $out/pytest_ffn.txt
if [ $rc -ne 0 ]; then grep -B5 -A25 "Error\|assert" $out/pytest_ffn.txt | head -80; exit 1; fi
timeout 600 python -m pytest tests/test_frame_gpu.py -x -q > $out/pytest_frame.txt 2>&1; echo "pytest frame rc $?"; tail -3 $out/pytest_frame.txt
for pc in 1 0 1 0; do
    MSSVT_FFN_PC=$pc ROWS=60 bash tools/prof.sh r04c_pc$pc --steps 20 > $out/prof_pc$pc.txt 2>&1
    echo "== MSSVT_FFN_PC=$pc"; grep "k_ffn_ws" $out/prof_pc$pc.txt | cut -d, -f1-4 | cut -c1-100
    MSSVT_FFN_PC=$pc timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $out/bench_pc$pc.json 2>> $out/bench.err
    python -c "
import json;d=json.loads([l for l in open('$out/bench_pc$pc.json') if l.startswith('{')][-1]);print('   bench', round(d['value'],1), round(d['ms_per_step'],4), d.get('timing',{}).get('median_ms'))"
done
for b in 8; do for pc in 1 0; do
    MSSVT_FFN_PC=$pc timeout 300 python bench.py --batch $b --no-cpu-baseline --no-roofline --steps 20 > $out/bench_b${b}_pc$pc.json 2>> $out/bench.err
    python -c "
import json;d=json.loads([l for l in open('$out/bench_b${b}_pc$pc.json') if l.startswith('{')][-1]);print('   bench b$b pc$pc', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
