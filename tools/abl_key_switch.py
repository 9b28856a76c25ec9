#!/usr/bin/env python3
"""tools/abl_key_switch.py — per-kernel times of 256 key switches (N=4096, k=1, l=61, base 2) through whatever build of
the library FHE_NTT_LIB points to; FHE_EXT32=0 selects the 61-bit kernels."""
import os, sys

os.environ.setdefault("FHE_NTT_ALLOW_ABLATED", "1")   # these tools load timing-only builds on purpose (binding.load_library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
B, L = pkg.binding, pkg.load_library()
n, k, l, batch, q = 4096, 1, 61, int(os.environ.get("KS_BATCH", "256")), pkg.Q61
plan = pkg.Plan(q, n)
rng = np.random.default_rng(3)
glwe = torch.from_numpy(rng.integers(0, q, (batch, k + 1, n), dtype=np.uint64).view(np.int64)).cuda()
ksk = torch.from_numpy(rng.integers(0, q, (k, l, k + 1, n), dtype=np.uint64).view(np.int64)).cuda()
out = torch.empty_like(glwe)
f = lambda: B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, glwe.data_ptr(), ksk.data_ptr(), out.data_ptr(), batch, 0, None))
for _ in range(3): f()
torch.cuda.synchronize()
B.kernel_timing_reset(); B.kernel_timing_enable(True)
for _ in range(10): f()
torch.cuda.synchronize()
t = {k: round(v[0] / v[1] * 1e3, 1) for k, v in B.kernel_timing_read().items()}
print(os.environ.get("FHE_EXT32", "1"), t, "sum", round(sum(t.values()), 1), "us")
