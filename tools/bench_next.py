#!/usr/bin/env python3
"""tools/bench_next.py — throughput of the next-row paths (BASELINE.json configs[2], configs[3])
on one GPU, with the oracle's schoolbook timed beside them.  Diagnostic (the contract bench is
bench.py); numbers are quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fhe_study_amd as pkg
from oracle import load_oracle

B, L, O = pkg.binding, pkg.load_library(), load_oracle()
st = torch.cuda.current_stream().cuda_stream
dev = "cuda:0"
U64 = 1 << 64


from _timing import timeit                           # warm clocks: tools/_timing.py


def bfv(n=8192, q=65537, t=2, batch=256):
    pq = q * q * q
    rng = np.random.default_rng(1)
    ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).to(dev)
    rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).to(dev)
    out = torch.empty((2, batch, n), dtype=torch.int64, device=dev)
    f = lambda: B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), out.data_ptr(), batch, st))
    dt = timeit(f)
    a = ab.cpu().numpy().view(np.uint64); r = rlk.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    w0, w1 = O.bfv_mul(q, n, t, pq, r[0], r[1], a[0, :1], a[1, :1], a[2, :1], a[3, :1])
    cpu = time.perf_counter() - t0
    ok = np.array_equal(out[0, 0].cpu().numpy().view(np.uint64), w0[0])
    print(f"BFV ct x ct mul + relin  N={n} q={q} p=q^2  batch={batch}: {dt*1e3:.2f} ms  "
          f"{batch/dt:,.0f} ct-mul/s   (oracle schoolbook, 1 core: {1/cpu:.2f} ct-mul/s)  parity={ok}")


def extprod(n=1024, k=1, l=64, batch=630, prepared=False):
    rng = np.random.default_rng(2)
    g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).to(dev)
    c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).to(dev)
    out = torch.empty_like(c)
    f = lambda: B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), out.data_ptr(), batch, st))
    if prepared:   # the key in its resident form (fhe_tggsw_prepare_dev once, then every product batch)
        prep = torch.empty(L.fhe_tggsw_prepared_words(n, k, l), dtype=torch.int64, device=dev)
        B._check(L.fhe_tggsw_prepare_dev(n, k, l, g.data_ptr(), prep.data_ptr(), st))
        f = lambda: B._check(L.fhe_tggsw_external_product_prepared_dev(n, k, l, prep.data_ptr(), c.data_ptr(), out.data_ptr(), batch, st))
    dt = timeit(f)
    t0 = time.perf_counter()
    w = O.external_product(n, k, l, g.cpu().numpy().view(np.uint64), c[:1].cpu().numpy().view(np.uint64))
    cpu = time.perf_counter() - t0
    ok = np.array_equal(out[0].cpu().numpy().view(np.uint64), w[0])
    print(f"TGGSW x TGLWE external product{' (prepared key)' if prepared else ''}  N={n} k={k} l={l}  batch={batch}: {dt*1e3:.3f} ms  "
          f"{batch/dt:,.0f} products/s   (oracle schoolbook, 1 core: {1/cpu:.2f}/s)  parity={ok}")


def key_switch(n=4096, k=1, beta=2, l=61, batch=256, q=pkg.Q61, resident_key=False, prepared=False):
    """GLWE<Rq>::key_switch (gfhe/src/glwe.rs:126-137): k*l*(k+1) products per ciphertext"""
    rng = np.random.default_rng(3)
    plan = pkg.Plan(q, n)
    glwe = torch.from_numpy(rng.integers(0, q, (batch, k + 1, n), dtype=np.int64)).to(dev)
    ksk = torch.from_numpy(rng.integers(0, q, (k, l, k + 1, n), dtype=np.int64)).to(dev)
    out = torch.empty_like(glwe)
    flags = 0
    key = ksk
    if resident_key:
        key = torch.empty_like(ksk)
        plan.forward_dev(ksk.data_ptr(), key.data_ptr(), k * l * (k + 1), st)
        flags = B.FHE_A_IS_EVALS
    f = lambda: B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, beta, l, glwe.data_ptr(), key.data_ptr(), out.data_ptr(), batch, flags, st))
    if prepared:
        prep = torch.empty(L.fhe_glwe_ksk_prepared_words(plan.handle, k, beta, l), dtype=torch.int64, device=dev)
        B._check(L.fhe_glwe_ksk_prepare_dev(plan.handle, k, beta, l, ksk.data_ptr(), prep.data_ptr(), st))
        f = lambda: B._check(L.fhe_glwe_key_switch_prepared_dev(plan.handle, k, beta, l, glwe.data_ptr(), prep.data_ptr(), out.data_ptr(), batch, st))
    dt = timeit(f)
    want = np.empty((k + 1, n), dtype=np.uint64)
    t0 = time.perf_counter()
    O.glue("key_switch", q, n, k, beta, l, glwe[0].cpu().numpy().view(np.uint64), ksk.cpu().numpy().view(np.uint64), want)
    cpu = time.perf_counter() - t0
    ok = np.array_equal(out[0].cpu().numpy().view(np.uint64), want)
    print(f"GLWE key switch  N={n} k={k} beta={beta} l={l} q~2^{q.bit_length()} batch={batch}"
          f"{' (key resident in NTT domain)' if resident_key else ' (prepared key)' if prepared else ''}: {dt*1e3:.3f} ms  {batch/dt:,.0f} switches/s   "
          f"(oracle, NTT products on 1 core: {1/cpu:.1f}/s)  parity={ok}")


if __name__ == "__main__":
    bfv()
    bfv(batch=2048)          # same work per ciphertext; launch gaps amortised
    extprod()
    extprod(prepared=True)
    key_switch()
    key_switch(prepared=True)
    key_switch(resident_key=True)
    key_switch(batch=1)
    key_switch(batch=1, prepared=True)
    key_switch(batch=1, resident_key=True)
    B.kernel_timing_reset(); B.kernel_timing_enable(True)
    extprod(batch=630, prepared=True)
    print({k: (round(v[0], 3), v[1]) for k, v in B.kernel_timing_read().items()})
    B.kernel_timing_reset()
    key_switch(prepared=True)
    print({k: (round(v[0], 3), v[1]) for k, v in B.kernel_timing_read().items()})
    B.kernel_timing_reset()
    key_switch(resident_key=True)
    print({k: (round(v[0], 3), v[1]) for k, v in B.kernel_timing_read().items()})
    B.kernel_timing_reset()
    bfv(batch=2048)
    print({k: (round(v[0], 3), v[1]) for k, v in B.kernel_timing_read().items()})
