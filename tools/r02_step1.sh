#!/bin/bash
# one GPU-box call: parity of the round-2 kernels, product / next-row throughput both ways, counters of the
# single-pass upper-bound experiment.  Output under gpurun_out/r2b/.
set -o pipefail
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] pytest -m gpu"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.txt
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/pytest_gpu.txt | head -40; exit 1; }
echo "[2] Rq product, fused middle vs round-1 chain"; date
timeout -k 10 200 python tools/mulbench.py 14:16384 15:8192 16:4096 17:2048 18:1024 13:16384 > $OUT/mulbench_fused.txt 2>&1 && cat $OUT/mulbench_fused.txt
FHE_RQ_MUL_FUSED=0 timeout -k 10 200 python tools/mulbench.py 14:16384 15:8192 16:4096 17:2048 18:1024 > $OUT/mulbench_unfused.txt 2>&1 && cat $OUT/mulbench_unfused.txt
echo "[3] next rows, fused digit-MAC vs round-1"; date
timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_fused.txt 2>&1 && cat $OUT/bench_next_fused.txt
FHE_DIGIT_MAC_FUSED=0 timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_unfused.txt 2>&1 && cat $OUT/bench_next_unfused.txt
echo "[4] counters of the single-pass upper-bound experiment"; date
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$tag -o run --output-format csv -- ./tools/ubench_fused 4096 > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
  rm -f $OUT/pmc_$tag/*/*.db
done
ls -R $OUT | head -40
date
