#!/usr/bin/env python3
"""tools/pmc_kernels.py <fetch_dir> <write_dir> [min_MB] — HBM-side bytes per launch of EVERY kernel of a run, from two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).  Counter units are KiB; on gfx950 FETCH_SIZE counts
64-byte requests as 32 bytes and is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import re, sys
from pmc_summary import per_kernel

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
for name in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    r, w = 2.0 * fetch.get(name, 0.0) * 1024 / 1e6, write.get(name, 0.0) * 1024 / 1e6
    if r + w < floor:
        continue
    short = re.sub(r"^void ", "", re.sub(r"\(.*\)$", "", name))
    print(f"{short:<72} read {r:10.2f} MB  write {w:10.2f} MB")
