#!/usr/bin/env python3
"""Fill the @PLACEHOLDER@ numbers of DESIGN.md / README.md from profiles/<tag>_* (tools/collect_profiles_r05.sh): python tools/fill_docs.py r05_b"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))  # noqa: E731
J = lambda n: json.load(open(P(n)))  # noqa: E731


def fps(v):
    s = "%d" % round(v)
    return s[:-3] + " " + s[-3:] if len(s) > 3 else s


b = J("bench_line.json")
one = J("bench_one_in_flight_line.json")
d20 = J("bench_driver_steps20_line.json") if os.path.exists(P("bench_driver_steps20_line.json")) else None
f32, f32one = J("bench_arith_f32_line.json"), J("bench_arith_f32_one_in_flight_line.json")
pts = J("bench_from_points_line.json")
b4, b8, b8bf, enl = J("bench_b4_f32_line.json"), J("bench_b8_f32_line.json"), J("bench_b8_bf16_line.json"), J("bench_enlarged_300k_line.json")
tr4, det = J("bench_train_b4_line.json"), J("bench_train_detector_b2_line.json")
rows = list(csv.DictReader(open(P("train_step_kernel_stats.csv"))))
tk = sum(float(r["TotalDurationNs"]) for r in rows) / 7e6
tl = sum(int(r["Calls"]) for r in rows) / 7
tt = re.search(r"compact path: ([0-9.]+) ms", open(P("train_time.txt")).read()).group(1)
r = b["roofline"]
odd, even = r["other_kernels"][0], r["other_kernels"][1]
v = {
    "B1_50": fps(b["value"]), "B1_50MS": "%.3f" % b["ms_per_step"],
    "B1_20": fps(d20["value"]) if d20 else "?", "B1_20MS": "%.3f" % d20["ms_per_step"] if d20 else "?",
    "B1_ONE": fps(one["value"]), "B1_ONEMS": "%.3f" % one["ms_per_step"],
    "F32": fps(f32["value"]), "F32MS": "%.3f" % f32["ms_per_step"], "F32_ONE": fps(f32one["value"]),
    "PTS": fps(pts["value"]), "PTSMS": "%.3f" % pts["ms_per_step"], "PTSFRAC": "%.3f" % pts["from_points"]["frac"],
    "PTS_ONE": fps((pts.get("one_frame_in_flight") or {}).get("value", 0)),
    "B4": fps(b4["value"]), "B4MS": "%.3f" % b4["ms_per_step"], "B8": fps(b8["value"]), "B8MS": "%.3f" % b8["ms_per_step"],
    "B8BF": fps(b8bf["value"]), "B8BFMS": "%.3f" % b8bf["ms_per_step"], "ENL": fps(enl["value"]), "ENLMS": "%.3f" % enl["ms_per_step"],
    "TR4": fps(tr4["value"]), "TR4MS": "%.1f" % tr4["ms_per_step"], "TR4SC": "%.1f" % (tr4["ms_per_step"] / 4),
    "TR1MS": tt, "TR1K": "%.1f" % tk, "TR1L": "%d" % round(tl),
    "DET": "%.1f" % det["value"], "DETMS": "%.0f" % det["ms_per_step"],
    "FFNFRAC": "%.3f" % r["frac"], "FFNISO": "%.1f" % r["ceiling"]["kernel_us_same_rows"], "FFNCEIL": "%.1f µs" % r["ceiling"]["us"],
    "FRAMEFRAC": "%.3f" % r["frame"]["frac"], "FRAMESUM": "%.3f" % r["frame"]["frac_of_sum"],
    "ATTNF16ODD": "%.2f" % odd["matrix"]["frac"], "ATTNF16EVEN": "%.2f" % even["matrix"]["frac"],
    "ATTNF32ODD": "%.2f" % odd["matrix"]["frac_vs_f32_matrix_peak"], "ATTNF32EVEN": "%.2f" % even["matrix"]["frac_vs_f32_matrix_peak"],
}
for name in ("DESIGN.md", "README.md"):
    path = os.path.join(ROOT, name)
    s = open(path).read()
    for k, val in v.items():
        s = s.replace("@%s@" % k, val)
    left = sorted(set(re.findall(r"@[A-Z0-9_]+@", s)))
    open(path, "w").write(s)
    print(name, "left:", left)
print(json.dumps(v, ensure_ascii=False))
