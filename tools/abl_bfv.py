#!/usr/bin/env python3
"""tools/abl_bfv.py — per-kernel times of 2048 BFV ciphertext products (N=8192, q=65537, p=q^2) through whatever build of the
library FHE_NTT_LIB points to (timing-only builds of tools/abl_build.sh give wrong words by design); FHE_EXT32=0: 61-bit kernels."""
import os, sys

os.environ.setdefault("FHE_NTT_ALLOW_ABLATED", "1")   # these tools load timing-only builds on purpose (binding.load_library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
B, L = pkg.binding, pkg.load_library()
n, q, t, batch = 8192, 65537, 2, int(os.environ.get("BFV_BATCH", "2048"))
pq = q * q * q
rng = np.random.default_rng(5)
ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).cuda()
rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).cuda()
out = torch.empty((2, batch, n), dtype=torch.int64, device="cuda")
f = lambda: B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), out.data_ptr(), batch, None))
from _timing import timeit
timeit(f)                                            # warm clocks (tools/_timing.py)
B.kernel_timing_reset(); B.kernel_timing_enable(True)
for _ in range(20): f()
torch.cuda.synchronize()
tm = {k: round(v[0] / 20 * 1e3, 1) for k, v in B.kernel_timing_read().items()}
print(os.path.basename(os.environ.get("FHE_NTT_LIB", "default")), tm, "sum", round(sum(tm.values()), 1), "us per step")
