#!/usr/bin/env python3
"""tools/kbench.py — kernel-level timing sweeps (diagnostic; not the contract bench).
usage: python tools/kbench.py [log_n] [batch] [tiles...]      (env KBENCH_Q: another modulus than 2^61 - 2^21 + 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_study_amd as pkg
from _timing import timeit

B = pkg.binding
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
tiles = [int(t) for t in sys.argv[3:]] or [0]
q, n = int(os.environ.get("KBENCH_Q", pkg.Q61)), 1 << log_n
plan = pkg.Plan(q, n)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
y = torch.empty_like(x)
B.fill_synthetic_dev(q, 1, 0, batch * n, x.data_ptr(), st)
for mode in ("fwd", "inv"):
    for tile in tiles:
        B.set_batch_tile(tile)
        f = (lambda: plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)) if mode == "fwd" else \
            (lambda: plan.inverse_dev(x.data_ptr(), y.data_ptr(), batch, st))
        dt = timeit(f)                               # warm clocks: tools/_timing.py
        B.kernel_timing_reset(); B.kernel_timing_enable(True)
        f(); torch.cuda.synchronize()
        kt = B.kernel_timing_read(); B.kernel_timing_enable(False)
        ks = " ".join(f"{k}:{ms/c*1e3:.1f}us x{c}" for k, (ms, c) in kt.items())
        print(f"{mode} n=2^{log_n} batch={batch} tile={tile}: {dt*1e3:.3f} ms  {batch/dt/1e6:.3f} M NTT/s  "
              f"{batch*n*16/dt/1e12:.2f} TB/s alg   [{ks}]", flush=True)
