#!/usr/bin/env python3
"""tools/pmc_summary.py — fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into
profiles/<round>_pmc_traffic.json, the file bench.py reads `roofline.traffic` from.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <polys_per_launch> <out.json>
Counter units are KiB; on gfx950 FETCH_SIZE counts 64-byte requests as 32 bytes and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, os, re, sys

NAMES = [  # (regex on the rocprofv3 kernel name, name used by the library's kernel timer)
    # the third template argument is the arithmetic (round 3: 0 / 1 / 2 = Shoup62 / Shoup61 / pseudo-Mersenne; `true` before)
    (r"ntt_fwd_strided_kernel<8, 32, (true|[012])(, false)?>", "ntt_fwd_strided_8"),
    (r"ntt_fwd_contig_kernel<8, true, (true|[012])(, 0)?>", "ntt_fwd_contig_final_8"),
    (r"ntt_inv_contig_kernel<8, false, false(, [012])?>", "ntt_inv_contig_8"),
    (r"ntt_inv_strided_kernel<8, 32(, [012])?>", "ntt_inv_strided_8"),
]


def short_name(rname):
    for pat, short in NAMES:
        m = re.search(pat, rname)
        if m:
            return short, m.group(0)
    return None, None


GRID = {}   # kernel name -> threads per launch (every pass kernel holds 16 coefficients per thread)


def per_kernel(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                GRID[name] = float(row.get("Grid_Size") or 0)
                key = (name, row["Dispatch_Id"])
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    for (name, _), v in acc.items():
        out.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    fetch_dir, write_dir, polys, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    n = 65536
    res = {"source": f"{os.path.relpath(out)}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                     f"`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (polynomials per launch from each launch's grid); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM",
           "polynomials_per_launch": polys, "kernels": {}}
    for rname, f_kb in fetch.items():
        short, matched = short_name(rname)
        if short is None or rname not in write:
            continue
        w_kb = write[rname]
        if GRID.get(rname):                  # polynomials per launch from the launch itself: the library tiles the batch
            polys = GRID[rname] * 16 / n
        hbm = (2.0 * f_kb + w_kb) * 1024.0
        res["kernels"][short] = {
            "rocprof_name": matched,
            "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
            "polynomials_per_launch": polys, "hbm_bytes_per_launch": hbm, "hbm_bytes_per_polynomial": hbm / polys,
            "algorithmic_bytes_per_polynomial": n * 16,
            "ratio_to_algorithmic": hbm / polys / (n * 16)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["kernels"], indent=1))
    if not res["kernels"]:
        sys.exit("pmc_summary: no kernel matched — update NAMES")


if __name__ == "__main__":
    main()
