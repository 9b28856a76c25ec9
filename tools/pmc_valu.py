#!/usr/bin/env python3
"""tools/pmc_valu.py <pmc_dir> <out.json> — per kernel of a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace`
run: average duration, effective clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 / duration), VALU instructions per
SIMD (SQ_INSTS_VALU counts wave instructions over the chip: / 1024 SIMDs) and cycles per VALU instruction per SIMD."""
import csv, glob, json, os, re, sys

d, out = sys.argv[1], sys.argv[2]
cnt, dur = {}, {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        key = (row["Kernel_Name"], row["Dispatch_Id"])
        cnt.setdefault(key, {}).setdefault(row["Counter_Name"], 0.0)
        cnt[key][row["Counter_Name"]] += float(row["Counter_Value"])
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[(row["Kernel_Name"], row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
res = {}
for (name, disp), c in cnt.items():
    if (name, disp) not in dur or "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_VALU" not in c:
        continue
    res.setdefault(name, []).append((dur[(name, disp)], c["GRBM_GUI_ACTIVE"], c["SQ_INSTS_VALU"]))
summary = {}
for name, rows in res.items():
    rows = rows[len(rows) // 2:]                                   # the later launches (warm)
    us = sum(r[0] for r in rows) / len(rows)
    gui = sum(r[1] for r in rows) / len(rows)
    valu = sum(r[2] for r in rows) / len(rows)
    if us < 5:
        continue
    clock = gui / 8 / (us * 1e-6) / 1e9
    per_simd = valu / 1024
    short = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))
    summary[short] = {"launches": len(rows), "avg_us": round(us, 1), "clock_GHz": round(clock, 3),
                      "valu_insts_per_simd": round(per_simd), "cycles_per_valu_inst": round(gui / 8 / per_simd, 2)}
json.dump(summary, open(out, "w"), indent=1)
for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"]):
    print(f"{k[:70]:<70} {v['avg_us']:9.1f} us  {v['clock_GHz']:.2f} GHz  {v['valu_insts_per_simd']:>9} VALU/SIMD  {v['cycles_per_valu_inst']:.2f} cyc/VALU")
