#!/bin/bash
# tools/mid_counters.sh — counters of the product's kernels at n = 2^16 (rq_mul_mid_kernel<8> and the strided passes):
# three rocprofv3 --pmc passes over tools/mulbench.py 16:4096, summed up per kernel by tools/pmc_isa.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b/midpmc; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/a -o run --output-format csv -- python3 $R/tools/mulbench.py 16:4096 > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $O/b -o run --output-format csv -- python3 $R/tools/mulbench.py 16:4096 > $O/b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/c -o run --output-format csv -- python3 $R/tools/mulbench.py 16:4096 > $O/c.log 2>&1
python3 $R/tools/pmc_isa.py $O/a $O/counters_a.json
python3 $R/tools/pmc_isa.py $O/b $O/counters_b.json
python3 $R/tools/pmc_isa.py $O/c $O/counters_c.json
tail -3 $O/c.log
