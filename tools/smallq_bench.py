#!/usr/bin/env python3
"""tools/smallq_bench.py — forward NTT and Rq x Rq at a small modulus (q = 65537) through the 32-bit kernels (default) or,
with FHE_EXT32=0, through the 61-bit ones: the A/B behind DESIGN.md's small-modulus paragraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
from _timing import timeit
rng = np.random.default_rng(1)
for q, n, batch in ((65537, 1024, 262144), (65537, 4096, 65536), (65537, 8192, 32768), (65537, 16384, 16384), (65537, 32768, 8192),
                    (786433, 65536, 4096), (786433, 131072, 2048)):
    plan = pkg.Plan(q, n)
    a = torch.from_numpy(rng.integers(0, q, (batch, n), dtype=np.int64)).cuda()
    b = torch.from_numpy(rng.integers(0, q, (batch, n), dtype=np.int64)).cuda()
    o = torch.empty_like(a)
    ops = [("forward", lambda: plan.forward_dev(a.data_ptr(), o.data_ptr(), batch)),
           ("inverse", lambda: plan.inverse_dev(a.data_ptr(), o.data_ptr(), batch))]
    ops.append(("Rq x Rq", lambda: plan.rq_mul_dev(a.data_ptr(), b.data_ptr(), o.data_ptr(), batch)))
    for name, f in ops:
        dt = timeit(f)                               # warm clocks: tools/_timing.py
        by = (16 if name != "Rq x Rq" else 24) * n * batch
        print(f"FHE_EXT32={os.environ.get('FHE_EXT32', '1')} q={q} n={n} batch={batch} {name:8s}: {dt*1e3:7.3f} ms  {batch/dt/1e6:8.2f} M/s  {by/dt/1e12:5.2f} TB/s algorithmic")
