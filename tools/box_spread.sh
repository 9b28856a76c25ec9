#!/bin/bash
# tools/box_spread.sh <tag> — one line per box of the pool: the contract bench and configs 2-4 on whatever box this call got
# (boxes differ by +-2 % on the headline).  Prints the line (and writes gpurun_out/spread_<tag>.txt on the box: gpurun merges
# that file back over the local one, so collect the printed lines):
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash tools/box_spread.sh r05' | tail -1 >> profiles/<tag>_box_spread_lines.txt
set -eo pipefail
TAG=${1:-r05}
OUT=gpurun_out/spread_$TAG.txt
mkdir -p gpurun_out
ID=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -1 || true)
line() { python3 - "$1" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        j=json.loads(ln); r=j.get("roofline",{})
        print(f"{j['value']:.0f} {j['unit']} ms/step {j['ms_per_step']:.3f} step_frac {r.get('step_frac',0):.4f}", end="")
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline > /tmp/b1.json 2>/dev/null
for c in 2 3 4; do timeout -k 10 200 python bench.py --config $c --no-cpu-baseline > /tmp/c$c.json 2>/dev/null; done
{ echo -n "$(date -u +%H:%M) gpu ${ID:-?} | headline: "; line /tmp/b1.json; for c in 2 3 4; do echo -n " | config $c: "; line /tmp/c$c.json; done; echo; } | tee -a $OUT
