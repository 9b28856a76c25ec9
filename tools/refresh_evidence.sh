#!/bin/bash
# tools/refresh_evidence.sh <tag> — regenerate the measured evidence in one GPU-box call:
#   /usr/local/graft/bin/gpurun --timeout 1190 -- 'bash tools/refresh_evidence.sh r03'
# Writes everything under gpurun_out/evid_<tag>/; the summaries the judge reads are copied into profiles/ by
# `bash tools/refresh_evidence.sh <tag> collect` run afterwards in the repo (no GPU needed).
# Steps are joined so that a failed or killed GPU step stops the script.
set -eo pipefail
TAG=${1:-r03}
OUT=gpurun_out/evid_$TAG
if [ "$2" = "collect" ]; then
  cp $OUT/bench.json profiles/${TAG}_bench.json
  cp $OUT/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
  cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
  cp $OUT/${TAG}_pmc_traffic.json profiles/
  for c in 2 3 4; do cp $OUT/config$c.json profiles/${TAG}_config$c.json; cp $OUT/config${c}_kernel_stats.csv profiles/${TAG}_config${c}_kernel_stats.csv; done
  cp $OUT/config4_pmc_traffic.txt profiles/${TAG}_config4_pmc_traffic.txt
  cp $OUT/pytest_gpu.txt profiles/${TAG}_pytest_gpu.txt
  cat $OUT/kbench_16.txt $OUT/kbench_12.txt > profiles/${TAG}_kbench.txt
  cp $OUT/bench_next_rows.txt profiles/${TAG}_bench_next_rows.txt
  cp $OUT/rq_mul_throughput.txt profiles/${TAG}_rq_mul_throughput.txt
  cp $OUT/ubench_bfly.txt profiles/${TAG}_ubench_bfly.txt
  cp $OUT/bfv_kernels.txt profiles/${TAG}_bfv_kernels.txt
  cp $OUT/config3_counters_a.json profiles/${TAG}_config3_valu_counters.json 2>/dev/null || true
  grep FHE_EXT32 $OUT/smallq.txt > profiles/${TAG}_small_modulus.txt
  cp $OUT/bench_2ranks.json profiles/${TAG}_bench_2ranks_one_gpu.json 2>/dev/null || true
  cp $OUT/bench_rccl_world1.json profiles/${TAG}_bench_rccl_world1.json 2>/dev/null || true
  cp $OUT/bench_pm0.json profiles/${TAG}_bench_shoup_kernels.json 2>/dev/null || true
  cp $OUT/isa_counters_a.json profiles/${TAG}_isa_counters_a.json 2>/dev/null || true
  cp $OUT/isa_counters_b.json profiles/${TAG}_isa_counters_b.json 2>/dev/null || true
  # the one-launch transform: one report per variant
  [ -s $OUT/persist_bench.txt ] && { python3 tools/persist_report.py $OUT ${TAG} || true; }
  for f in bench_persist_A bench_persist_B bench_persist_E bench_rank_of_8 bench_mg bench_mg_shoup bench_self_launch; do [ -s $OUT/$f.json ] && grep '^{' $OUT/$f.json | tail -1 > profiles/${TAG}_$f.json; done
  [ -s $OUT/single_pass_bound.txt ] && { cat tools/single_pass_bound_reading.txt $OUT/single_pass_bound.txt > profiles/${TAG}_single_pass_bound.txt; }
  # JSON evidence files hold the JSON line only (RCCL prints banners on stdout)
  for f in profiles/${TAG}_bench_rccl_world1.json profiles/${TAG}_bench_2ranks_one_gpu.json; do [ -s $f ] && { grep '^{' $f | tail -1 > $f.tmp; mv $f.tmp $f; }; done
  sed -i '/amdgpu.ids/d' profiles/${TAG}_*.txt
  exit 0
fi
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity"

echo "[1/7] GPU parity tests"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -30 $OUT/pytest_gpu.txt; exit 1; }
tail -2 $OUT/pytest_gpu.txt

echo "[2/7] PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)"; date
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o runc --output-format csv -- python3 $B > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o runc --output-format csv -- python3 $B > $OUT/pmc_write.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write 65536 $OUT/${TAG}_pmc_traffic.json > $OUT/pmc_summary.txt
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json     # bench.py reads it from profiles/

echo "[2b] dynamic instruction counters of the pass kernels (beside profiles/<tag>_isa_*.txt)"; date
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_isa_a -o runc --output-format csv -- python3 $B > $OUT/pmc_isa_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/pmc_isa_b -o runc --output-format csv -- python3 $B > $OUT/pmc_isa_b.log 2>&1
python3 tools/pmc_isa.py $OUT/pmc_isa_a $OUT/isa_counters_a.json | grep -E "ntt_(fwd|inv)" || true
python3 tools/pmc_isa.py $OUT/pmc_isa_b $OUT/isa_counters_b.json | grep -E "ntt_(fwd|inv)" || true

echo "[3/7] bench.py (contract run)"; date
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json

FHE_PM=0 timeout -k 10 600 python bench.py --no-cpu-baseline > $OUT/bench_pm0.json 2> $OUT/bench_pm0.err    # the Shoup kernels on the same box
python3 -c "
import json
a=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1]); b=json.loads([l for l in open('$OUT/bench_pm0.json') if l.startswith('{')][-1])
print('pseudo-Mersenne', round(a['value']), 'NTT/s step_frac', round(a['roofline']['step_frac'],4), '| Shoup kernels (FHE_PM=0)', round(b['value']), round(b['roofline']['step_frac'],4))"

echo "[4/7] rocprofv3 --kernel-trace --stats of the same command"; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o runc --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
head -4 $OUT/bench_kernel_stats.csv

echo "[5/7] configs 2-4: bench line + kernel stats of the same command"; date
for c in 2 3 4; do
  timeout -k 10 300 python bench.py --config $c --cpu-seconds 8 > $OUT/config$c.json 2> $OUT/config$c.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_c$c -o runc --output-format csv -- python3 bench.py --config $c --no-cpu-baseline > $OUT/config${c}_under_rocprof.json 2> $OUT/rocprof_c$c.err
  cp $(find $OUT/stats_c$c -name "*kernel_stats.csv" | head -1) $OUT/config${c}_kernel_stats.csv
  python3 -c "
import json,sys
o=json.loads([l for l in open('$OUT/config$c.json') if l.startswith('{')][-1])
print('config $c:', round(o['value'],1), o['unit'], 'ms/step', round(o['ms_per_step'],3), 'hbm frac', round(o['roofline']['frac'],4), 'valu frac', round(o['roofline']['valu']['frac'],3), 'cpu', round(o['cpu_baseline']['value'],1))"
done

echo "[5b] HBM-side traffic of config 4's kernels (PMC, separate passes)"; date
C4="bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-parity"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc4_fetch -o runc --output-format csv -- python3 $C4 > $OUT/pmc4_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc4_write -o runc --output-format csv -- python3 $C4 > $OUT/pmc4_write.log 2>&1
{ echo "# HBM-side traffic per launch of the config-4 kernels (630 TGGSW x TGLWE, N=1024, k=1, l=64; plain entry point)"
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, FETCH_SIZE doubled on gfx950) -- python3 $C4"
  echo "# algorithmic: ciphertexts in 10.3 MB, key 2.1 MB in / 4.2 MB of transforms (x 8 XCD L2s), partial sums 630 x 4 parts x 32 KiB = 82.6 MB out and in, result 10.3 MB"
  python3 tools/pmc_kernels.py $OUT/pmc4_fetch $OUT/pmc4_write; } > $OUT/config4_pmc_traffic.txt
cat $OUT/config4_pmc_traffic.txt

echo "[6/7] two ranks on one GPU (gloo rendezvous): the multi-rank driver"; date
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --share-gpu --backend gloo --global-batch 4096 --no-cpu-baseline --parity-all-ranks --gather-check --allgather > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err || tail -5 $OUT/bench_2ranks.err
tail -1 $OUT/bench_2ranks.json | cut -c1-300

echo "[6b] ONE rank through RCCL (world size 1): process group, barrier, MAX all-reduce, all-gather of the real shard"; date
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 3 --warmup 1 --backend nccl --force-dist --allgather --batch-per-gpu 2048 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err || tail -5 $OUT/bench_rccl_world1.err
tail -1 $OUT/bench_rccl_world1.json | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print(o.get('distributed'), o.get('allgather'))" || true

echo "[7/7] inverse / other sizes / product / next rows (diagnostic)"; date
timeout -k 10 300 python tools/kbench.py 16 16384 0 > $OUT/kbench_16.txt 2>&1
timeout -k 10 300 python tools/kbench.py 12 262144 0 > $OUT/kbench_12.txt 2>&1
timeout -k 10 300 python tools/mulbench.py > $OUT/rq_mul_throughput.txt 2>&1
timeout -k 10 300 python tools/bench_next.py > $OUT/bench_next_rows.txt 2>&1 || true
[ -x tools/ubench_bfly ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_bfly tools/ubench_bfly.hip
timeout -k 10 200 ./tools/ubench_bfly > $OUT/ubench_bfly.txt 2>&1 || true
timeout -k 10 200 python tools/smallq_bench.py > $OUT/smallq.txt 2>&1 || true
FHE_EXT32=0 timeout -k 10 200 python tools/smallq_bench.py >> $OUT/smallq.txt 2>&1 || true
timeout -k 10 100 python tools/abl_bfv.py > $OUT/bfv_kernels.txt 2>&1 || true
FHE_EXT32=0 timeout -k 10 100 python tools/abl_bfv.py >> $OUT/bfv_kernels.txt 2>&1 || true
# 2^62 <= q < 2^63 (generic63.hip) beside the headline modulus on this box
(for ln in 16 12 10; do KBENCH_Q=9223372036844421121 timeout -k 10 120 python tools/kbench.py $ln $((1 << (28 - ln))); done; timeout -k 10 120 python tools/kbench.py 16 4096) > $OUT/q63_kbench.txt 2>&1 || true
# VALU counters of the BFV kernels (two --pmc passes, as for the pass kernels)
C3="bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc3_a -o runc --output-format csv -- python3 $C3 > $OUT/pmc3_a.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc3_b -o runc --output-format csv -- python3 $C3 > $OUT/pmc3_b.log 2>&1 || true
python3 tools/pmc_isa.py $OUT/pmc3_a $OUT/config3_counters_a.json > /dev/null 2>&1 || true
python3 tools/pmc_isa.py $OUT/pmc3_b $OUT/config3_counters_b.json > /dev/null 2>&1 || true
cat $OUT/kbench_16.txt $OUT/kbench_12.txt

# single-HBM-pass upper bounds on the production butterflies (tools/ubench_fused.hip) and the passes without their butterflies
[ -x tools/ubench_fused ] && { timeout -k 10 300 ./tools/ubench_fused 8192 > $OUT/single_pass_bound.txt 2>&1 || true; }
# q = 1 (mod 2^32): the word-Montgomery kernels against the Shoup ones on this box (NOT BASELINE's modulus); one rank of 8
QMG=2305842979148922881
timeout -k 10 300 python bench.py --q $QMG --no-cpu-baseline > $OUT/bench_mg.json 2> $OUT/bench_mg.err || true
FHE_MG=0 timeout -k 10 300 python bench.py --q $QMG --no-cpu-baseline > $OUT/bench_mg_shoup.json 2> $OUT/bench_mg_shoup.err || true
timeout -k 10 300 python bench.py --global-batch 8192 --steps 40 --no-cpu-baseline > $OUT/bench_rank_of_8.json 2> $OUT/bench_rank_of_8.err || true
# `python bench.py --gpus 2` WITHOUT a launcher (it starts its own ranks; both share this GPU, gloo rendezvous)
env -u RANK -u WORLD_SIZE -u LOCAL_RANK timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --share-gpu --backend gloo --global-batch 4096 --no-cpu-baseline --parity-all-ranks > $OUT/bench_self_launch.json 2> $OUT/bench_self_launch.err || tail -5 $OUT/bench_self_launch.err
if [ "${EVID_PERSIST:-0}" != "1" ]; then
  for f in bench_rank_of_8 bench_mg bench_mg_shoup bench_self_launch; do python3 -c "
import json
o=json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][-1]); print('$f', round(o['value']), o['unit'], 'n_gpus', o['n_gpus'], 'ms/step', round(o['ms_per_step'],3), 'step_frac', round(o['roofline']['step_frac'],4))" || true; done
  find $OUT -name "*.db" -delete 2>/dev/null || true
  date
  exit 0
fi
echo "[8] (EVID_PERSIST=1) the one-launch transform (variants A and B): ms per step at three settings each, per-part profile, traffic, counters"; date
PERSIST_PROFILE=1 timeout -k 10 900 python tools/persist_bench.py 65536 A:16,1,0 A:64,1,0 A:256,1,0 B:1,1 B:2,1 B:4,1 D:1,1 D:2,1 E:2,1 E:3,1 E:4,1 E:6,1 > $OUT/persist_bench.txt 2>&1 || tail -5 $OUT/persist_bench.txt
grep -E "^time|parity:" $OUT/persist_bench.txt || true
for cfg in two-pass A:64,1,0 B:2,1 D:1,1 E:4,1; do
  tag=$(echo $cfg | tr ':,' '__')
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pt_f_$tag -o run --output-format csv -- python3 tools/persist_one.py $cfg 8192 3 > $OUT/pt_f_$tag.log 2>&1 || true
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pt_w_$tag -o run --output-format csv -- python3 tools/persist_one.py $cfg 8192 3 > $OUT/pt_w_$tag.log 2>&1 || true
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pt_a_$tag -o run --output-format csv -- python3 tools/persist_one.py $cfg 8192 3 > $OUT/pt_a_$tag.log 2>&1 || true
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/pt_b_$tag -o run --output-format csv -- python3 tools/persist_one.py $cfg 8192 3 > $OUT/pt_b_$tag.log 2>&1 || true
  { echo "== $cfg: HBM-side bytes per launch (8192 polynomials per launch = 4295 MB in + 4295 MB out; two-pass: per 2048-polynomial launch = 1074 + 1074 MB per pass)"
    python3 tools/pmc_kernels.py $OUT/pt_f_$tag $OUT/pt_w_$tag 100
    echo "== $cfg: instruction counters per wave"
    python3 tools/pmc_isa.py $OUT/pt_a_$tag $OUT/persist_counters_a_$tag.json | grep -v fill_synthetic
    python3 tools/pmc_isa.py $OUT/pt_b_$tag $OUT/persist_counters_b_$tag.json | grep -v fill_synthetic; } >> $OUT/persist_counters.txt 2>&1 || true
done
cat $OUT/persist_counters.txt || true
FHE_NTT_PERSIST=A:64,1,0 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_persist_A.json 2> $OUT/bench_persist_A.err || true
FHE_NTT_PERSIST=B:2,1 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_persist_B.json 2> $OUT/bench_persist_B.err || true
FHE_NTT_PERSIST=E:4,1 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_persist_E.json 2> $OUT/bench_persist_E.err || true
# q = 1 (mod 2^32): the word-Montgomery forward kernels against the Shoup ones on this box (NOT BASELINE's modulus)
QMG=2305842979148922881
timeout -k 10 300 python bench.py --q $QMG --no-cpu-baseline > $OUT/bench_mg.json 2> $OUT/bench_mg.err || true
FHE_MG=0 timeout -k 10 300 python bench.py --q $QMG --no-cpu-baseline > $OUT/bench_mg_shoup.json 2> $OUT/bench_mg_shoup.err || true
timeout -k 10 300 python bench.py --global-batch 8192 --steps 40 --no-cpu-baseline > $OUT/bench_rank_of_8.json 2> $OUT/bench_rank_of_8.err || true
for f in bench_persist_A bench_persist_B bench_persist_E bench_rank_of_8 bench_mg bench_mg_shoup; do python3 -c "
import json
o=json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][-1]); print('$f', round(o['value']), o['unit'], 'ms/step', round(o['ms_per_step'],3), 'step_frac', round(o['roofline']['step_frac'],4))" || true; done
rm -rf $OUT/pmc3_*/*/*.db $OUT/pmc_isa_*/*/*.db $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db $OUT/pmc4_*/*/*.db $OUT/stats*/*/*.db 2>/dev/null || true
rm -rf $OUT/pt_*/*/*.db 2>/dev/null || true
find $OUT -name "*.db" -delete 2>/dev/null || true
date
