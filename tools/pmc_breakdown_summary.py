"""Per-kernel mean of every counter of tools/pmc_breakdown.sh's passes -> one JSON, plus derived shares of the wave cycles.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md): shares below are
fractions of SQ_WAVE_CYCLES; per-wave-instruction figures divide by the instruction counts."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    fs = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if name.startswith("_Z"):
            m = re.match(r"_Z\d+(k_\w+?)I((?:L[ib]\d+E)+)Ev", name)
            if m:
                args = [("true" if a[2:] == "1" else "false") if a[1] == "b" else a[2:] for a in m.group(2).split("E") if a]
                name = m.group(1) + "<" + ", ".join(args) + ">"
        if name.startswith("k_"):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {c: int(v) for c, v in m.items()}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if c in m:
                e["share_" + c[3:].lower()] = round(m[c] / wc, 3)
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
top = sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]
for k, e in top:
    print(k, {x: e[x] for x in sorted(e) if x.startswith("share_")},
          {x: e.get(x) for x in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU",
                                 "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM",
                                 "SQ_INST_CYCLES_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")})
