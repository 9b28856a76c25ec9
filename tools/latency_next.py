#!/usr/bin/env python3
"""tools/latency_next.py — time per CALL of the prepared external product (N=1024, k=1, l=64) and the prepared key switch
(N=4096, l=61) at small batches: what a blind rotation (630 dependent external products per bootstrap, batched over the
bootstraps in flight) sees.  Device-resident, back-to-back calls on one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
B, L = pkg.binding, pkg.load_library()
rng = np.random.default_rng(9)

def timeit(f, reps=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

n, k, l = 1024, 1, 64
g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
prep = torch.empty(L.fhe_tggsw_prepared_words(n, k, l), dtype=torch.int64, device="cuda")
B._check(L.fhe_tggsw_prepare_dev(n, k, l, g.data_ptr(), prep.data_ptr(), None))
for batch in (1, 8, 64, 256, 630):
    c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()
    o = torch.empty_like(c)
    dt = timeit(lambda: B._check(L.fhe_tggsw_external_product_prepared_dev(n, k, l, prep.data_ptr(), c.data_ptr(), o.data_ptr(), batch, None)))
    print(f"external product (prepared key) N={n} batch={batch:4d}: {dt*1e6:8.1f} us per call  {batch/dt:12,.0f} products/s")
q, n, l = pkg.Q61, 4096, 61
plan = pkg.Plan(q, n)
ksk = torch.from_numpy(rng.integers(0, q, (k, l, k + 1, n), dtype=np.int64)).cuda()
kp = torch.empty(L.fhe_glwe_ksk_prepared_words(plan.handle, k, 2, l), dtype=torch.int64, device="cuda")
B._check(L.fhe_glwe_ksk_prepare_dev(plan.handle, k, 2, l, ksk.data_ptr(), kp.data_ptr(), None))
for batch in (1, 8, 64, 256):
    glwe = torch.from_numpy(rng.integers(0, q, (batch, k + 1, n), dtype=np.int64)).cuda()
    o = torch.empty_like(glwe)
    dt = timeit(lambda: B._check(L.fhe_glwe_key_switch_prepared_dev(plan.handle, k, 2, l, glwe.data_ptr(), kp.data_ptr(), o.data_ptr(), batch, None)))
    print(f"key switch (prepared key) N={n} batch={batch:4d}: {dt*1e6:8.1f} us per call  {batch/dt:12,.0f} switches/s")
