#!/usr/bin/env python3
"""tools/persist_bench.py — the one-launch n = 2^16 forward transform (csrc/ntt_persist.hip) against the two-pass kernels:
word-for-word comparison on small / ragged batches and from 1 / 7 / 20 / 100 workgroups, then timings over a list of settings
(PERSIST_PROFILE=1: lane 0's shader-clock ticks per part of an iteration as well).
usage: python tools/persist_bench.py [batch] [A:T,L,R | B:R[,s] | D:R[,s] | E:R[,s] ...]       (diagnostic; not the contract bench)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fhe_study_amd as pkg

B = pkg.binding
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
def parse(a):
    kind, rest = a.split(":")
    vals = [int(v) for v in rest.split(",")]
    return (1, vals[0], vals[1], vals[2]) if kind.upper() == "A" else ({"B": 2, "D": 3, "E": 4}[kind.upper()], 1, vals[1] if len(vals) > 1 else 0, vals[0])


cfgs = [parse(a) for a in sys.argv[2:]] or [parse("B:2,1"), parse("B:4,1"), parse("A:16,1,0"), parse("A:64,1,0")]
name = lambda c: f"A:{c[1]},{c[2]},{c[3]}" if c[0] == 1 else f"B:{c[3]},{c[2]}" if c[0] == 2 else f"D:{c[3]},{c[2]}" if c[0] == 3 else f"E:{c[3]},{c[2]}" if c[0] == 4 else "two-pass"
q, n = pkg.Q61, 1 << 16
plan = pkg.Plan(q, n)
st = torch.cuda.current_stream().cuda_stream


def run(x, y, b):
    plan.forward_dev(x.data_ptr(), y.data_ptr(), b, st)


# ---- parity: every setting against the two-pass kernels, batches that leave tiles, items and queues ragged ----
bad = 0
for b in (1, 3, 37, 300, 1030):
    x = torch.empty(b * n, dtype=torch.int64, device="cuda:0")
    B.fill_synthetic_dev(q, 7 + b, 0, b * n, x.data_ptr(), st)
    ref = torch.empty_like(x)
    B.set_persist(0)
    run(x, ref, b)
    torch.cuda.synchronize()
    for cfg in cfgs:
        y = torch.zeros_like(x)
        B.set_persist(*cfg)
        t0 = time.perf_counter()
        run(x, y, b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            B.persist_status()
            err = ""
        except Exception as e:   # noqa: BLE001
            err = f" STATUS: {e}"
        ok = bool(torch.equal(y, ref))
        z = x.clone()                      # in place
        run(z, z, b)
        torch.cuda.synchronize()
        ok2 = bool(torch.equal(z, ref))
        bad += (not ok) + (not ok2) + bool(err)
        print(f"parity batch={b:5d} {name(cfg)}: {'ok' if ok else 'MISMATCH'} / in place {'ok' if ok2 else 'MISMATCH'}"
              f"  ({dt*1e3:.2f} ms){err}", flush=True)
    del x, ref
# ---- no co-residency is assumed: the same words from 1, 7, 20 and 100 workgroups ----
for grid in (1, 7, 20, 100):
    B.set_persist_grid(grid)
    b = 21
    x = torch.empty(b * n, dtype=torch.int64, device="cuda:0")
    B.fill_synthetic_dev(q, 99, 0, b * n, x.data_ptr(), st)
    ref = torch.empty_like(x)
    B.set_persist(0)
    run(x, ref, b)
    torch.cuda.synchronize()
    for cfg in cfgs:
        y = torch.zeros_like(x)
        B.set_persist(*cfg)
        t0 = time.perf_counter()
        run(x, y, b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            B.persist_status()
            err = ""
        except Exception as e:   # noqa: BLE001
            err = f" STATUS: {e}"
        ok = bool(torch.equal(y, ref))
        # a queue nobody serves (fewer workgroups than XCDs) must be REPORTED by the teams, never a silent hole
        expected_err = cfg[0] >= 2 and grid < 8
        good = (ok and not err) or (expected_err and "never served" in err)
        bad += not good
        print(f"grid={grid:4d} batch={b} {name(cfg)}: {'ok' if ok else 'MISMATCH'} ({dt*1e3:.2f} ms){err}{'' if good else '  <-- FAILURE'}", flush=True)
B.set_persist_grid(0)
print("parity:", "ALL OK" if not bad else f"{bad} FAILURES", flush=True)
if bad and not os.environ.get("PERSIST_BENCH_FORCE"):
    sys.exit(1)

# ---- timings ----
x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
y = torch.empty_like(x)
B.fill_synthetic_dev(q, 1, 0, batch * n, x.data_ptr(), st)
for cfg in [(0, 0, 0, 0)] + cfgs:
    T = cfg[0]
    B.set_persist(*cfg)
    for _ in range(2):
        run(x, y, batch)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run(x, y, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    B.persist_status()
    prof = ""
    if T and os.environ.get("PERSIST_PROFILE"):
        pw = torch.zeros(26, dtype=torch.int64, device="cuda:0")
        B.load_library().fhe_ntt_persist_profile(pw.data_ptr())
        run(x, y, batch)
        torch.cuda.synchronize()
        B.load_library().fhe_ntt_persist_profile(None)
        w = pw.cpu().tolist()
        names = ["top-barrier", "poll/late-loads", "half1", "look-ahead", "xchg-barrier", "half2-rest", "hand-over", "gather", "round1", "epi-a", "next-loads", "store-barrier"]
        for ph, nm in ((0, "S"), (1, "C")):
            items = max(w[24 + ph], 1)
            tot = sum(w[ph * 12:ph * 12 + 12])
            prof += f"\n      {nm}: {items} items, {tot / items:.0f} ticks/item: " + ", ".join(f"{n} {w[ph * 12 + i] / items:.2f}" if 0 < w[ph * 12 + i] / items < 10 else f"{n} {w[ph * 12 + i] / items:.0f}" for i, n in enumerate(names))
    print(f"time batch={batch} {name(cfg)}: {dt*1e3:.3f} ms  {batch/dt/1e6:.3f} M NTT/s  {batch*n*16/dt/8e12:.4f} of 8 TB/s"
          f"  (x 65536/batch = {dt*1e3*65536/batch:.2f} ms per step){prof}", flush=True)
