#!/bin/bash
# tools/refresh_evidence.sh <tag> — regenerate the measured evidence in one GPU-box call:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/refresh_evidence.sh r01'
# Writes everything under gpurun_out/evid_<tag>/; copy the summaries into profiles/ afterwards
# (tools/refresh_evidence.sh prints the cp commands).  Steps are joined so that a failed or
# killed GPU step stops the script.
set -eo pipefail
TAG=${1:-r01}
OUT=gpurun_out/evid_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity"

echo "[1/6] GPU parity tests"; date
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -30 $OUT/pytest_gpu.txt; exit 1; }
tail -2 $OUT/pytest_gpu.txt

echo "[2/6] PMC pass FETCH_SIZE"; date
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o runc --output-format csv -- python3 $B > $OUT/pmc_fetch.log 2>&1
echo "[3/6] PMC pass WRITE_SIZE"; date
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o runc --output-format csv -- python3 $B > $OUT/pmc_write.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write 65536 profiles/${TAG}_pmc_traffic.json > $OUT/pmc_summary.txt
cp profiles/${TAG}_pmc_traffic.json $OUT/

echo "[4/6] bench.py (contract run)"; date
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json

echo "[5/6] rocprofv3 --kernel-trace --stats of the same command"; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o runc --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
head -4 $OUT/bench_kernel_stats.csv

echo "[6/6] inverse / other sizes / next rows (diagnostic)"; date
timeout -k 10 300 python tools/kbench.py 16 16384 0 > $OUT/kbench_16.txt 2>&1
timeout -k 10 300 python tools/kbench.py 12 262144 0 > $OUT/kbench_12.txt 2>&1
timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_rows.txt 2>&1 || true
cat $OUT/kbench_16.txt $OUT/kbench_12.txt
rm -rf $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db 2>/dev/null || true
echo "copy into profiles/:"
echo "  cp $OUT/bench.json profiles/${TAG}_bench.json; cp $OUT/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv"
echo "  cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json; cp $OUT/${TAG}_pmc_traffic.json profiles/"
