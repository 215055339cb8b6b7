"""Sweep of tests/test_module_gpu.py::test_random_configurations_match_the_oracle over a seed range (GPU box):

    python tests/sweep_oracle.py <first seed> <count>

Not collected by pytest; the committed test runs six of these seeds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_module_gpu as t  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    for impl in ("fused", "ops"):
        try:
            t.test_random_configurations_match_the_oracle(impl, seed)
            print(seed, impl, "ok", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(seed, impl, "FAIL", type(e).__name__, str(e).splitlines()[0][:120] if str(e) else "", flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
