#!/bin/bash
# on the GPU box: bench.py --config 3 with each library of tools/ab_bfv_r04.sh (three runs each, the production library first and last)
cd "$(dirname "$0")/.."
run() { for i in 1 2 3; do FHE_NTT_LIB=$1 timeout -k 10 120 python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), end=' ')"; done; echo; }
echo -n "production (T=2,R=1): "; run ""
for f in fhe-study_amd/build/abl/libfhe_ntt_bfv_*.so; do echo -n "$(basename $f .so | sed s/libfhe_ntt_bfv_//): "; run $PWD/$f; done
echo -n "production again: "; run ""
