"""Per-kernel means of the counters of tools/pmc_frame.sh's passes -> one JSON (the file fused.roofline / DESIGN.md cite).

hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane
read and is doubled (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included in both.  cycles = GRBM_GUI_ACTIVE / 8
XCDs; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles; valu_issue = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs / cycles.
Counter runs serialise kernels: durations are longer than in the kernel-trace profiles."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if name.startswith("_Z"):  # mangled (kernels with vector-typed arguments): _Z8k_ffn_wsILi128ELi256ELb1ELb1EEv... 
            m = re.match(r"_Z\d+(k_\w+?)I((?:L[ib]\d+E)+)Ev", name)
            if m:
                args = [("true" if a[2:] == "1" else "false") if a[1] == "b" else a[2:] for a in m.group(2).split("E") if a]
                name = m.group(1) + "<" + ", ".join(args) + ">"
        if not name.startswith("k_"):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_how": __doc__}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches_seen": len(next(iter(cs.values())))}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in m:
            e[c + "_KB"] = round(m[c], 1)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        e["cycles_per_launch"] = int(cyc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 3)
        if "SQ_INSTS_VALU" in m:
            e["valu_issue_frac"] = round(4.0 * m["SQ_INSTS_VALU"] / 1024.0 / cyc, 3)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA"):
        if c in m:
            e[c] = int(m[c])
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m and m["SQ_WAVE_CYCLES"] > 0:
        e["wave_wait_frac"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3)
    # per query pattern for the attention launches (alternating heavy / light): min and max of the fetch size
    if "FETCH_SIZE" in cs and len(cs["FETCH_SIZE"]) > 1:
        e["FETCH_SIZE_KB_min_max"] = [round(min(cs["FETCH_SIZE"]), 1), round(max(cs["FETCH_SIZE"]), 1)]
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "_how"})[:3000])
