#!/bin/bash
# GPU-box call 3 of round 2: parity, fused digit-MAC (scaled workgroups, fused tail) vs unfused.
set -o pipefail
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] pytest -m gpu"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/pytest_gpu.txt | head -40; exit 1; }
echo "[2] next rows: variants"; date
timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_default.txt 2>&1; grep -v amdgpu.ids $OUT/bench_next_default.txt
echo "== FHE_DIGIT_MAC_FUSED=0"
FHE_DIGIT_MAC_FUSED=0 timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_unfused.txt 2>&1; grep -v amdgpu.ids $OUT/bench_next_unfused.txt
date
