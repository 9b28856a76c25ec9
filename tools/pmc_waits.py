#!/usr/bin/env python3
"""tools/pmc_waits.py <pmc_dir> — per kernel of a `rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace` run: where a wave's cycles go
(MI355X_MICROARCH.md: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing;
the three are disjoint and sum to WAVE_CYCLES; all in quad-cycles)."""
import csv, glob, json, os, re, sys

d = sys.argv[1]
cnt, dur = {}, {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        key = (row["Kernel_Name"], row["Dispatch_Id"])
        c = cnt.setdefault(key, {})
        c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[(row["Kernel_Name"], row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
by = {}
for key, c in cnt.items():
    if key in dur and c.get("SQ_WAVES"):
        by.setdefault(key[0], []).append((dur[key], c))
for name, rows in sorted(by.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
    rows = rows[len(rows) // 2:]
    us = sum(r[0] for r in rows) / len(rows)
    if us < 20:
        continue
    avg = {k: sum(r[1].get(k, 0.0) for r in rows) / len(rows) for k in rows[0][1]}
    w = avg["SQ_WAVES"]
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    short = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))[:70]
    out = {"avg_us": round(us, 1), "waves": round(w), "wave_cycles_per_wave(quad)": round(wc / w)}
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if k in avg and wc:
            out[k + "_frac_of_wave_cycles"] = round(avg[k] / wc, 3)
    print(short, json.dumps(out))
