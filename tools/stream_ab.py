#!/usr/bin/env python3
"""tools/stream_ab.py — the same device-resident calls on the legacy NULL stream and on an explicit (non-blocking) stream:
forward NTT N = 2^16 x 16384 (8 launch pairs), N = 4096 x 4096, and 2048 BFV products.  (Round 5: no difference once the
chip is warm; a first version with 3 warm-up calls showed 5 % for whichever came first.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
B, L = pkg.binding, pkg.load_library()
side = torch.cuda.Stream()
from _timing import timeit
def bench(name, f, reps):
    return timeit(f)
for log_n, batch, reps in ((16, 16384, 10), (12, 4096, 200)):
    n = 1 << log_n
    plan = pkg.Plan(pkg.Q61, n)
    x = torch.empty(batch * n, dtype=torch.int64, device="cuda"); y = torch.empty_like(x)
    B.fill_synthetic_dev(pkg.Q61, 1, 0, batch * n, x.data_ptr(), None)
    for name, st in (("NULL", None), ("explicit", side.cuda_stream)):
        dt = bench(name, lambda: plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st), reps)
        print(f"fwd n=2^{log_n} batch={batch} {name:9s}: {dt*1e3:8.4f} ms  {batch/dt/1e6:.3f} M NTT/s", flush=True)
    del x, y
n, q, t, batch = 8192, 65537, 2, 2048
pq = q * q * q
rng = np.random.default_rng(5)
rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).cuda()
ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).cuda()
out = torch.empty((2, batch, n), dtype=torch.int64, device="cuda")
for name, st in (("explicit", side.cuda_stream), ("NULL", None), ("explicit", side.cuda_stream), ("NULL", None)):   # (order swapped and repeated: warm-up is not the difference)
    dt = bench(name, lambda: B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), out.data_ptr(), batch, st)), 10)
    print(f"bfv 2048 pairs {name:9s}: {dt*1e3:7.3f} ms  {batch/dt/1e6:.3f} M ct-mul/s", flush=True)
