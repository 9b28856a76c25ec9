#!/usr/bin/env python3
"""tools/mulbench.py — throughput of the device-resident Rq product (fhe_rq_mul_dev) per size;
FHE_RQ_MUL_FUSED=0 selects the three-kernel path for single-pass sizes (diagnostic); MULBENCH_Q: another modulus than
2^61 - 2^21 + 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_study_amd as pkg
from _timing import timeit

B = pkg.binding
st = torch.cuda.current_stream().cuda_stream
sizes = [(4, 1 << 20), (8, 1 << 18), (10, 1 << 17), (12, 1 << 15), (13, 1 << 14), (14, 1 << 14), (16, 4096), (18, 1024)] + [(10, 1), (12, 1), (13, 1)]
if len(sys.argv) > 1:   # tools/mulbench.py 14:16384 16:4096 ...
    sizes = [tuple(int(x) for x in arg.split(":")) for arg in sys.argv[1:]]
print(f"FHE_RQ_MUL_FUSED={os.environ.get('FHE_RQ_MUL_FUSED', '1')}", flush=True)
for log_n, batch in sizes:
    q, n = int(os.environ.get("MULBENCH_Q", pkg.Q61)), 1 << log_n
    plan = pkg.Plan(q, n)
    a = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    b = torch.empty_like(a); c = torch.empty_like(a); ce = torch.empty_like(a)
    B.fill_synthetic_dev(q, 1, 0, batch * n, a.data_ptr(), st)
    B.fill_synthetic_dev(q, 2, 0, batch * n, b.data_ptr(), st)
    work = torch.empty(max(16, plan.workspace_bytes(batch) // 8), dtype=torch.int64, device="cuda:0")
    f = lambda: plan.rq_mul_dev(a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, d_c_evals=ce.data_ptr(), d_work=work.data_ptr(), stream=st)
    dt = timeit(f)                                   # warm clocks: tools/_timing.py
    B.kernel_timing_reset(); B.kernel_timing_enable(True)
    f(); torch.cuda.synchronize()
    kt = B.kernel_timing_read(); B.kernel_timing_enable(False)
    ks = " ".join(f"{k}:{ms/c*1e3:.1f}us" + (f"x{c}" if c > 1 else "") for k, (ms, c) in kt.items())
    print(f"rq_mul n=2^{log_n} batch={batch}: {dt*1e6:10.1f} us  {batch/dt/1e6:8.3f} M products/s  "
          f"{batch*n*24/dt/1e12:.2f} TB/s alg (24n)  [{ks}]", flush=True)
