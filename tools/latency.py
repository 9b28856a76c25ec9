#!/usr/bin/env python3
"""tools/latency.py — single-call latency of the HOST-buffer entry points (what a shim that swaps
the bodies of NTT::ntt / Rq mul would pay per call), batch = 1 and small batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fhe_study_amd as pkg

rng = np.random.default_rng(1)
for q, n in [(65537, 4), (65537, 512), (pkg.Q61, 1024), (pkg.Q61, 8192), (pkg.Q61, 65536)]:
    plan = pkg.Plan(q, n)
    for batch in (1, 16):
        a = rng.integers(0, q, (batch, n), dtype=np.uint64)
        b = rng.integers(0, q, (batch, n), dtype=np.uint64)
        for name, f in (("ntt", lambda: plan.forward(a)), ("rq_mul", lambda: plan.rq_mul(a, b, want_evals=False))):
            for _ in range(3):
                f()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            dt = (time.perf_counter() - t0) / reps
            print(f"q~2^{q.bit_length()} n={n:6d} batch={batch:3d} {name:7s} {dt*1e6:9.1f} us/call", flush=True)

# ---- device-resident chain of small products: direct launches vs one captured hipGraph ----------
import torch
q, n, batch, links = pkg.Q61, 1024, 1, 8
plan = pkg.Plan(q, n)
x = torch.from_numpy(rng.integers(0, q, (links + 1, batch * n), dtype=np.int64)).cuda()
y = torch.from_numpy(rng.integers(0, q, batch * n, dtype=np.int64)).cuda()
work = torch.empty(plan.workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")


def chain(st):
    for i in range(links):   # x[i+1] = x[i] * y
        plan.rq_mul_dev(x[i].data_ptr(), y.data_ptr(), x[i + 1].data_ptr(), batch, d_work=work.data_ptr(), stream=st)


def timed(f, reps=200):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


direct = timed(lambda: chain(torch.cuda.current_stream().cuda_stream))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain(torch.cuda.current_stream().cuda_stream)
graph = timed(g.replay)
print(f"chain of {links} Rq products (n={n}, device-resident, one fused kernel each): direct {direct*1e6:.1f} us, "
      f"captured hipGraph {graph*1e6:.1f} us per chain")
