#!/usr/bin/env python3
"""tools/kstats.py <kernel_stats.csv> — print a rocprofv3 --stats kernel table compactly."""
import csv, signal, sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)   # `| head` is fine
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>4s} total_ms={int(r['TotalDurationNs'])/1e6:8.3f} "
          f"avg_us={float(r['AverageNs'])/1e3:9.1f}")
