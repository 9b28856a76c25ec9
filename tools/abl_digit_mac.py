#!/usr/bin/env python3
"""tools/abl_digit_mac.py — kernel times of 630 external products (N=1024) through whatever build of the library
FHE_NTT_LIB points to.  Used with the timing-only builds of digit_mac.hip (-DFHE_DM_ABLATE_MAC / -DFHE_DM_ABLATE_NTT:
the fused kernel without its multiply phase / without its transform) to split the kernel's time; numbers in digit_mac.hip."""
import os, sys

os.environ.setdefault("FHE_NTT_ALLOW_ABLATED", "1")   # these tools load timing-only builds on purpose (binding.load_library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
B, L = pkg.binding, pkg.load_library()
n, k, l, batch = int(os.environ.get("EXT_N", "1024")), 1, 64, int(os.environ.get("EXT_BATCH", "630"))
rng = np.random.default_rng(2)
g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()
out = torch.empty_like(c)
prep = torch.empty(L.fhe_tggsw_prepared_words(n, k, l), dtype=torch.int64, device="cuda")
B._check(L.fhe_tggsw_prepare_dev(n, k, l, g.data_ptr(), prep.data_ptr(), None))
f = lambda: B._check(L.fhe_tggsw_external_product_prepared_dev(n, k, l, prep.data_ptr(), c.data_ptr(), out.data_ptr(), batch, None))
from _timing import timeit
timeit(f)                                            # warm clocks (tools/_timing.py)
B.kernel_timing_reset(); B.kernel_timing_enable(True)
for _ in range(10): f()
torch.cuda.synchronize()
print(os.environ.get("FHE_NTT_LIB", "default"), {k: round(v[0] / v[1] * 1e3, 1) for k, v in B.kernel_timing_read().items()})
