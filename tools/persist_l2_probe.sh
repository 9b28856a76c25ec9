#!/bin/bash
# tools/persist_l2_probe.sh — does the teams' intermediate stay in the L2?  HBM-side reads / writes of variant B (8192 polynomials
# per launch: 4295 MB in, 4295 MB out, 4295 MB written to and read back from the ring) at 1, 2 and 4 workgroups per CU:
# fewer workgroups = fewer polynomials between "written" and "read" per XCD.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c/l2probe; mkdir -p $O
for grid in 256 512 1024; do
  for cfg in B:2,1 B:1,1; do
    tag=$(echo ${cfg}_g$grid | tr ':,' '__')
    FHE_NTT_PERSIST_GRID=$grid timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f_$tag -o run --output-format csv -- python3 $R/tools/persist_one.py $cfg 8192 3 > $O/f_$tag.log 2>&1
    FHE_NTT_PERSIST_GRID=$grid timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w_$tag -o run --output-format csv -- python3 $R/tools/persist_one.py $cfg 8192 3 > $O/w_$tag.log 2>&1
    FHE_NTT_PERSIST_GRID=$grid timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s_$tag -o run --output-format csv -- python3 $R/tools/persist_one.py $cfg 8192 3 > $O/s_$tag.log 2>&1
    us=$(python3 - <<PY
import csv,glob
rows=[r for p in glob.glob("$O/s_$tag/**/*kernel_stats.csv",recursive=True) for r in csv.DictReader(open(p)) if "team" in r["Name"]]
print(round(float(rows[0]["AverageNs"])/1e3,1) if rows else "?")
PY
)
    echo "== $cfg, $grid workgroups ($((grid/256)) per CU): kernel ${us} us"
    python3 $R/tools/pmc_kernels.py $O/f_$tag $O/w_$tag 100 | grep team
  done
done
