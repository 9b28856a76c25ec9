#!/bin/bash
# GPU-box call 2 of round 2: parity after the digit-table round 0, A/B of the fused digit-MAC variants.
set -o pipefail
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] pytest -m gpu"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/pytest_gpu.txt | head -40; exit 1; }
echo "[2] next rows: variants"; date
for v in "default" "FHE_DIGIT_MAC_WAVES=2" "FHE_DIGIT_MAC_MAXACC=32" "FHE_DIGIT_MAC_FUSED=0"; do
  echo "== $v"
  if [ "$v" = "default" ]; then timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_default.txt 2>&1; cat $OUT/bench_next_default.txt | grep -v amdgpu.ids
  else env_name=${v%%=*}; timeout -k 10 300 env $v python tools/bench_next.py > $OUT/bench_next_$env_name.txt 2>&1; grep -v amdgpu.ids $OUT/bench_next_$env_name.txt; fi
done
echo "[3] product"; date
timeout -k 10 200 python tools/mulbench.py 14:16384 16:4096 18:1024 > $OUT/mulbench_fused.txt 2>&1 && grep -v amdgpu.ids $OUT/mulbench_fused.txt
date
