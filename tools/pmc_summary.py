"""Per-kernel mean of every counter in a rocprofv3 --pmc output directory."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(t in k for t in ("k_ffn", "k_attn", "k_window_plan", "k_plan_order", "k_cmp", "k_segment", "k_nms")):
        continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
