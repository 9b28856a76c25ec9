#!/usr/bin/env python3
"""tools/persist_one.py CFG [batch] [reps] — run the n = 2^16 forward transform under one setting (two-pass | A:T,L,R | B:R,s)
a few times and nothing else: the program to put behind `rocprofv3 --pmc ... --` (tools/pmc_kernels.py sums it up)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_study_amd as pkg

B = pkg.binding
cfg = sys.argv[1] if len(sys.argv) > 1 else "two-pass"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
q, n = pkg.Q61, 1 << 16
plan = pkg.Plan(q, n)
st = torch.cuda.current_stream().cuda_stream
if cfg != "two-pass":
    kind, rest = cfg.split(":")
    v = [int(t) for t in rest.split(",")]
    B.set_persist(1, v[0], v[1], v[2]) if kind.upper() == "A" else B.set_persist({"B": 2, "D": 3, "E": 4}[kind.upper()], 1, v[1] if len(v) > 1 else 0, v[0])
x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
y = torch.empty_like(x)
B.fill_synthetic_dev(q, 1, 0, batch * n, x.data_ptr(), st)
for _ in range(reps):
    plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)
torch.cuda.synchronize()
B.persist_status()
print("done", cfg, batch, reps)
