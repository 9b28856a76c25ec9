#!/usr/bin/env python3
"""tools/pmc_isa.py <pmc_dir> <out.json> — per kernel of a
`rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace`
run: the dynamic instruction counts PER WAVE (counter / SQ_WAVES) that stand beside the static histograms of
tools/isa_hist.py in profiles/r03_isa_*.txt, the effective clock (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration)
and cycles per VALU instruction per SIMD."""
import csv, glob, json, os, re, sys

d, out = sys.argv[1], sys.argv[2]
cnt, dur = {}, {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        key = (row["Kernel_Name"], row["Dispatch_Id"])
        c = cnt.setdefault(key, {})
        c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[(row["Kernel_Name"], row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
by_kernel = {}
for key, c in cnt.items():
    if key in dur and c.get("SQ_WAVES"):
        by_kernel.setdefault(key[0], []).append((dur[key], c))
summary = {}
for name, rows in by_kernel.items():
    rows = rows[len(rows) // 2:]                       # the later launches (warm)
    us = sum(r[0] for r in rows) / len(rows)
    if us < 5:
        continue
    avg = {k: sum(r[1].get(k, 0.0) for r in rows) / len(rows) for k in rows[0][1]}
    waves = avg["SQ_WAVES"]
    short = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))
    e = {"launches": len(rows), "avg_us": round(us, 1), "waves": round(waves)}
    for k, v in avg.items():
        if k.startswith("SQ_INSTS") or k.startswith("SQ_WAIT"):
            e[k + "_per_wave"] = round(v / waves, 1)
    if "GRBM_GUI_ACTIVE" in avg:
        e["clock_GHz"] = round(avg["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9, 3)
        if "SQ_INSTS_VALU" in avg:
            e["cycles_per_valu_inst_per_simd"] = round(avg["GRBM_GUI_ACTIVE"] / 8 / (avg["SQ_INSTS_VALU"] / 1024), 2)
    summary[short] = e
json.dump(summary, open(out, "w"), indent=1)
for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"]):
    print(k[:90], json.dumps(v))
