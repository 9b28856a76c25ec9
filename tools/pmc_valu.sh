#!/bin/bash
# tools/pmc_valu.sh — clock and VALU issue counters of the config-3 and config-4 kernels (one GPU-box call):
#   gpurun --timeout 600 -- 'bash tools/pmc_valu.sh'   ->  gpurun_out/pmc_valu/config{3,4}_valu.json (+ .txt)
set -eo pipefail
OUT=gpurun_out/pmc_valu; mkdir -p $OUT; export TMPDIR=/tmp
for c in 3 4; do
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace -d $OUT/c$c -o run --output-format csv -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c$c.log 2>&1
  python3 tools/pmc_valu.py $OUT/c$c $OUT/config${c}_valu.json | tee $OUT/config${c}_valu.txt
  find $OUT/c$c -name "*.db" -delete 2>/dev/null || true
done
