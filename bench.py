#!/usr/bin/env python3
"""bench.py — throughput of the hot path on MI355X with the roofline and the CPU baseline in the
same JSON line.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

Default workload = BASELINE.json configs[4] (`--config 5`): forward negacyclic NTT, N = 65536,
q = 2^61-2^21+1, a GLOBAL batch of 65536 polynomials (32 GiB in + 32 GiB out), synthetic coefficients
generated on the device, device-resident in -> out.  A "step" is one pass of the
hot path over the batch.  With N ranks the 65536 polynomials are block-partitioned by fhe_shard_range
(8192 per GPU, 4 GiB, at N = 8 — what configs[4] and SURVEY.md §8d state): total work is fixed as N
grows, "scaling": "strong"; at N = 1 that is the whole batch on one GPU.  `--batch-per-gpu B` selects
the weak variant instead (B polynomials on every rank; `config.workload` says which).  There is no
data-path collective; `--allgather` times the one optional collective — every rank's
REAL shard gathered onto every rank through sharding.all_gather_rows — on its own, never in `value`.

`--config {2,3,4}` measure the other BASELINE.json configurations with the same JSON shape
(their own metric names; the contract line is the default):
  2  forward + inverse NTT, N = 4096, batch 4096                     -> NTT/s
  3  BFV ciphertext x ciphertext multiply + relinearise, N = 8192     -> ct-mul/s
  4  TGGSW x TGLWE external product, N = 1024, k = 1, l = 64, 630 products (per GPU: the
     fhe_shard_range share of 630 when N > 1)                         -> products/s

`roofline` describes the WHOLE step against the HBM roofline (algorithmic bytes of SURVEY.md §8d
per unit / measured time per unit); `roofline.kernels` holds every kernel's own average launch
time (HIP events on the launch stream) and, for the dominant one, its own algorithmic rate;
`roofline.valu` prices the same step against the measured integer ceiling of the chip
(butterflies per second of tools/ubench_bfly.hip), which is what binds every configuration here.
PyTorch is used for device memory, streams and torch.distributed only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q61 = 2305843009211596801
Q16 = 65537
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Chip-wide ceiling of the 64-bit Shoup/Harvey butterfly in registers, no memory at all:
# tools/ubench_bfly.hip v8 (the production form) at 8 waves per SIMD, profiles/r01_ubench_bfly.txt.
VALU_PEAK_GBFLY = 2018.7
# the five-multiply butterfly of pseudo-Mersenne moduli (zq_device.hpp: ct_bfly_pm), which the headline modulus runs:
# tools/ubench_bfly.hip v18 (13 instructions, round 4) at 8 waves per SIMD, profiles/r04_ubench_bfly.txt
VALU_PEAK_GBFLY_PM = 2837.2
# the word-Montgomery butterfly of moduli q = 1 (mod 2^32) (zq_device.hpp: ct_bfly_mg / gs_bfly_mg: transforms and products): tools/ubench_bfly.hip
# v14 at 8 waves per SIMD, profiles/r04_ubench_bfly.txt (v14 is the statement-per-instruction form of it: a lower bound of the peak)
VALU_PEAK_GBFLY_MG = 2219.9
# the 32-bit butterfly of the small-prime kernels (digit32.hip / bfv32.hip): tools/ubench_bfly.hip v13, registers only
VALU_PEAK_GBFLY32 = 5730.0
# HBM traffic per launch of each kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
# WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM
# prescribes for gfx950).  Counters cannot be read from inside this process.
PMC_TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: per config, see CONFIG_STEPS)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: per config)")
    ap.add_argument("--config", type=int, default=5, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration (1-based; 5 = the headline, the contract line)")
    ap.add_argument("--log-n", type=int, default=None)
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="weak scaling: this many units on EVERY rank (config 5 default is the strong form below)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="configs 4 and 5: shard exactly this many units over the ranks with fhe_shard_range (strong "
                         "scaling).  Config 5 defaults to 65536 = configs[4]; config 4: 630 = configs[3]")
    ap.add_argument("--q", type=int, default=None)
    ap.add_argument("--batch-tile", type=int, default=0, help="polynomials per launch (0 = library default)")
    ap.add_argument("--prewarm", type=float, default=0.15,
                    help="seconds of untimed steps BEFORE the --warmup steps, so that the chip is at its sustained clocks (0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-all-ranks", action="store_true",
                    help="every rank checks its own shard against the oracle (default: rank 0 only)")
    ap.add_argument("--gather-check", action="store_true",
                    help="config 5, N > 1: all-gather the shards (sharding.ShardedNTT driven by Plan.forward_dev) "
                         "and compare with the single-rank transform of the whole batch on rank 0")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--allgather", action="store_true",
                    help="config 5: also time the all-gather of every rank's result shard onto every rank "
                         "(sharding.all_gather_rows; RCCL with --backend nccl), reported separately, never in `value`")
    ap.add_argument("--allgather-chunk-rows", type=int, default=0,
                    help="gather in chunks of this many rows per rank (0 = ONE collective)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even at WORLD_SIZE 1 (needs the launcher's env): the whole N > 1 "
                         "code path — process group, barrier, MAX all-reduce, all-gather — through RCCL on one GPU")
    return ap.parse_args()


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, 16)   # the GPU box's CPU share for one GPU


def threaded_rate(fn, units_per_call, seconds, cores):
    """fn() processes `units_per_call` units on one thread (ctypes releases the GIL): run it on
    `cores` threads for about `seconds`; returns (units/s, units, wall seconds)."""
    import threading

    t0 = time.perf_counter()
    fn()
    per_call = time.perf_counter() - t0
    calls = max(1, int(seconds / per_call))
    done = [0] * cores

    def work(i):
        for _ in range(calls):
            fn()
            done[i] += units_per_call

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return sum(done) / dt, sum(done), dt, units_per_call / per_call


# ------------------------------------------------------------------------------------------------
# workloads: each returns a dict of closures / constants over device-resident buffers
# ------------------------------------------------------------------------------------------------
def workload_ntt(args, pkg, torch, dev, st, rank, world, inverse_too):
    B = pkg.binding
    if inverse_too:   # config 2
        q, log_n, batch = args.q or Q61, args.log_n or 12, args.batch_per_gpu or 4096
        seed = 0xF4E50002
        total, b0 = world * batch, rank * batch
        strong = False
    else:             # config 5
        q, log_n = args.q or Q61, args.log_n or 16
        seed = 0xF4E50005
        if args.batch_per_gpu and not args.global_batch:     # weak variant: the same batch on every rank
            batch, strong = args.batch_per_gpu, False
            total, b0 = world * batch, rank * batch
        else:                                                # configs[4]: ONE global batch, block-partitioned
            total, strong = args.global_batch or 65536, True
            b0, b1 = B.shard_range(total, world, rank)
            batch = b1 - b0
    n = 1 << log_n
    plan = pkg.Plan(q, n)
    B._check(pkg.load_library().fhe_ntt_plan_prepare(plan.handle))
    x = torch.empty(batch * n, dtype=torch.int64, device=dev)
    y = torch.empty(batch * n, dtype=torch.int64, device=dev)
    z = torch.empty(batch * n, dtype=torch.int64, device=dev) if inverse_too else None
    first = b0 * n                # this rank owns rows [b0, b0 + batch) of the global batch
    B.fill_synthetic_dev(q, seed, first, batch * n, x.data_ptr(), st)

    def step():
        plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)
        if inverse_too:
            plan.inverse_dev(y.data_ptr(), z.data_ptr(), batch, st)

    def parity(O, np):
        rng = np.random.default_rng(7)
        rows = sorted(set(list(range(min(8, batch))) + list(range(max(0, batch - 8), batch)) +
                          [int(r) for r in rng.integers(0, batch, 48)])) if batch else []
        Y = y.view(batch, n)
        bad = 0
        for r in rows:
            a = O.fill_synthetic(q, seed, first + r * n, n)
            if not np.array_equal(Y[r].cpu().numpy().view(np.uint64), O.ntt(q, n, a)):
                bad += 1
        if inverse_too and not torch.equal(z, x):
            bad += 1
        return len(rows), bad

    def cpu(O, np, seconds):
        """The oracle's reference-cost-model port (16-byte {q,v} AoS, u128 %, per-call table clone under
        a mutex — arith/src/ntt.rs:20-73) on this host's cores, on a bounded sample of the same rows."""
        O.roots(q, n)  # table build is a one-off in the reference too (CACHE): keep it out of the timing
        cores = host_cores()
        probe = O.fill_synthetic(q, seed, 0, 4 * n)
        t0 = time.perf_counter()
        O.ref_ntt_aos(q, n, probe, threads=1)
        per_ntt = (time.perf_counter() - t0) / 4
        count = max(cores, int(seconds / per_ntt) // cores * cores)
        count = min(count, 8192 if n >= 16384 else 1 << 18)
        xs = O.fill_synthetic(q, seed, 0, count * n)
        t0 = time.perf_counter()
        O.ref_ntt_aos(q, n, xs, threads=cores)
        dt = time.perf_counter() - t0
        return {"value": count / dt, "unit": "NTT/s", "cores": cores, "kind": "port",
                "sample": f"{count} forward NTTs of the same synthetic rows (N={n}), {cores} threads, {dt:.1f} s; "
                          f"single thread: {1.0 / per_ntt:.1f} NTT/s",
                "value_1core": 1.0 / per_ntt}

    transforms = 2 if inverse_too else 1
    arith = plan.arithmetic()                              # which exact Zq::mul the kernels run (fhe_ntt_plan_arithmetic)
    small_q = arith == 3                                   # FHE_ARITH_WORD32 of include/fhe_ntt.h (the PUBLIC numbering): smallq.hip, 32-bit words (NOT the BASELINE modulus: --q given)
    if inverse_too:
        name = (f"batched forward+inverse negacyclic NTT, N={n}, q={q}, {batch} polynomials per GPU "
                "(BASELINE.json configs[1] shape), device-resident")
    elif strong:
        name = (f"batched forward negacyclic NTT, N={n}, q={q}, global batch {total} polynomials block-partitioned over "
                f"{world} GPU(s) by fhe_shard_range ({batch} on rank 0; BASELINE.json configs[4]; strong scaling), "
                "device-resident in->out")
    else:
        name = (f"batched forward negacyclic NTT, N={n}, q={q}, {batch} polynomials on EVERY GPU (--batch-per-gpu: the WEAK "
                f"variant of BASELINE.json configs[4], global batch {total}), device-resident in->out")
    metric = ("NTT/s (N=4096, 64-bit q, forward+inverse) per node; achieved HBM GB/s vs roofline" if inverse_too
              else "NTT/s (N=2^16, 64-bit q) per node; achieved HBM GB/s vs roofline")
    if args.q is not None and q != Q61 and not small_q:
        # a modulus given with --q: the shape is the config's, the modulus (and with it the arithmetic the kernels run) is not
        name += f" — q given with --q ({pkg.Plan.ARITH_NAMES.get(arith, arith)} arithmetic), NOT BASELINE.json's modulus 2^61 - 2^21 + 1"
    if small_q:
        metric = (f"NTT/s (N=2^{log_n}, q={q}: a {q.bit_length()}-bit modulus given with --q, NOT BASELINE.json's 64-bit q) per node; "
                  "achieved HBM GB/s vs roofline")
    return {
        "metric": metric,
        "unit": "NTT/s", "units_per_step": transforms * batch, "step": step, "parity": parity, "cpu": cpu,
        "alg_bytes_per_unit": 16 * n,            # read N + write N coefficients of 8 B
        "bfly_per_unit": (n // 2) * log_n,
        "bfly_bits": 32 if small_q else 64, "dtype": "u32" if small_q else "u64", "arith": arith,
        # every pass kernel reads and writes each coefficient once (the small-modulus two-pass sizes keep u32 in between)
        "pass_bytes_per_launch_per_unit": 12 * n if (small_q and log_n > 14) else 16 * n,
        "config": {"workload": name, "n": n, "q": q, "batch_per_gpu": batch, "global_batch": total,
                   "arithmetic": pkg.Plan.ARITH_NAMES[arith],
                   "parallelism": f"batch-sharded x{world}, no collective"},
        "global_units_per_step": transforms * total, "scaling": "strong" if strong else "weak",
        "plan": plan, "x": x, "y": y, "n": n, "q": q, "batch": batch, "seed": seed, "total": total, "b0": b0,
    }


def workload_bfv(args, pkg, torch, dev, st, rank, world):
    """config 3: RLWE::mul = tensor + relinearize_204 (bfv/src/lib.rs:59-90,251-271) at N = 8192 with the
    parameters the reference's i64/f64 arithmetic admits: q = 65537, t = 2, p = q^2."""
    import numpy as np

    B, L = pkg.binding, pkg.load_library()
    q, n, t = Q16, 1 << (args.log_n or 13), 2
    batch = args.batch_per_gpu or 2048
    pq = q * q * q
    rng = np.random.default_rng(0xF4E50003 + rank)
    ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).to(dev)
    rlk = torch.from_numpy(np.random.default_rng(0xF4E50003).integers(0, pq, (2, n), dtype=np.int64)).to(dev)
    out = torch.empty((2, batch, n), dtype=torch.int64, device=dev)

    def step():
        B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), out.data_ptr(), batch, st))

    def parity(O, np):
        a, r = ab.cpu().numpy().view(np.uint64), rlk.cpu().numpy().view(np.uint64)
        rows, bad = [0, batch - 1], 0
        for i in rows:
            w0, w1 = O.bfv_mul(q, n, t, pq, r[0], r[1], a[0, i:i + 1], a[1, i:i + 1], a[2, i:i + 1], a[3, i:i + 1])
            if not (np.array_equal(out[0, i].cpu().numpy().view(np.uint64), w0[0]) and
                    np.array_equal(out[1, i].cpu().numpy().view(np.uint64), w1[0])):
                bad += 1
        return len(rows), bad

    def cpu(O, np, seconds):
        a, r = ab[:, :1].cpu().numpy().view(np.uint64), rlk.cpu().numpy().view(np.uint64)
        cores = host_cores()
        fn = lambda: O.bfv_mul(q, n, t, pq, r[0], r[1], a[0], a[1], a[2], a[3])
        rate, units, dt, one = threaded_rate(fn, 1, seconds, cores)
        return {"value": rate, "unit": "ct-mul/s", "cores": cores, "kind": "port",
                "sample": f"{units} ciphertext products (the reference's schoolbook ring_n::naive_mul + f64 scale-round, "
                          f"oracle/fhe_next_oracle.c) on {cores} threads, {dt:.1f} s; single thread {one:.2f}/s",
                "value_1core": one}

    ext32 = os.environ.get("FHE_EXT32", "1")[:1] != "0"      # FHE_EXT32=0: the general 61-bit path (zring.hip)
    log2n = (2 * n).bit_length() - 1
    # bfv32.hip (small q): tensor 2 primes x (4 forward + 3 inverse) size-2N transforms, relinearisation 3 primes x
    # (1 forward + 2 inverse), all 32-bit butterflies (the key's forward transforms are per call, not per ciphertext)
    transforms = 2 * 7 + 3 * 3
    return {
        "metric": "BFV ct x ct multiply + relinearise per second (N=8192, q=65537, t=2, p=q^2) per node",
        "unit": "ct-mul/s", "units_per_step": batch, "step": step, "parity": parity, "cpu": cpu,
        "alg_bytes_per_unit": (4 + 2) * 8 * n,      # two ciphertexts in, one out; the key is shared by the batch
        "bfly_per_unit": (transforms if ext32 else 13) * n * log2n,    # (2N/2) * log2(2N) per size-2N transform; 61-bit path: 13 of them
        "bfly_bits": 32 if ext32 else 64,
        "pass_bytes_per_launch_per_unit": None,
        "config": {"workload": f"RLWE::mul (tensor + relinearize_204), N={n}, q={q}, t={t}, p=q^2, {batch} ciphertext pairs "
                               "per GPU (BASELINE.json configs[2]), device-resident",
                   "n": n, "q": q, "batch_per_gpu": batch, "global_batch": world * batch,
                   "parallelism": f"batch-sharded x{world}, no collective"},
    }


def workload_extprod(args, pkg, torch, dev, st, rank, world):
    """config 4: 630 TGGSW x TGLWE external products (tfhe/src/tggsw.rs:45-62), N = 1024, k = 1, l = 64;
    with N ranks each owns its fhe_shard_range block of the 630 (79 x 7 + 77 at 8)."""
    import numpy as np

    B, L = pkg.binding, pkg.load_library()
    n, k, l = 1 << (args.log_n or 10), 1, 64
    # weak scaling by default (630 products per GPU); --global-batch 630 block-shards exactly 630 as configs[3] states
    total = args.global_batch if args.global_batch else (args.batch_per_gpu or 630) * world
    b0, b1 = B.shard_range(total, world, rank)
    batch = b1 - b0
    g = torch.from_numpy(np.random.default_rng(0xF4E50004).integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).to(dev)
    allc = np.random.default_rng(0xF4E50104).integers(-(1 << 63), 1 << 63, (total, k + 1, n), dtype=np.int64)
    c = torch.from_numpy(allc[b0:b1].copy()).to(dev)
    out = torch.empty_like(c)

    def step():
        B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), out.data_ptr(), batch, st))

    def parity(O, np):
        rows = sorted(set([0, batch // 2, batch - 1])) if batch else []
        if not rows:
            return 0, 0
        want = O.external_product(n, k, l, g.cpu().numpy().view(np.uint64), c[rows].cpu().numpy().view(np.uint64))
        bad = sum(0 if np.array_equal(out[r].cpu().numpy().view(np.uint64), want[i]) else 1 for i, r in enumerate(rows))
        return len(rows), bad

    def cpu(O, np, seconds):
        gh, ch = g.cpu().numpy().view(np.uint64), c[:1].cpu().numpy().view(np.uint64)
        cores = host_cores()
        rate, units, dt, one = threaded_rate(lambda: O.external_product(n, k, l, gh, ch), 1, seconds, cores)
        return {"value": rate, "unit": "products/s", "cores": cores, "kind": "port",
                "sample": f"{units} external products (the reference's schoolbook Tn x Tn, oracle/fhe_next_oracle.c) on "
                          f"{cores} threads, {dt:.1f} s; single thread {one:.2f}/s",
                "value_1core": one}

    ext32 = os.environ.get("FHE_EXT32", "1")[:1] != "0"      # FHE_EXT32=0: the 61-bit fused kernels (digit_mac.hip), one prime
    log2n = n.bit_length() - 1
    # digit32.hip: every transform modulo two 27-bit primes (32-bit butterflies)
    transforms = (2 if ext32 else 1) * ((k + 1) * l + 2 * (k + 1))     # digit transforms + the inverses of the two 32-bit key halves
    return {
        "metric": "TGGSW x TGLWE external products per second (N=1024, k=1, l=64) per node",
        "unit": "products/s", "units_per_step": batch, "step": step, "parity": parity, "cpu": cpu,
        "alg_bytes_per_unit": 2 * (k + 1) * 8 * n,    # ciphertext in + out; the TGGSW key is shared by the batch
        "bfly_per_unit": transforms * (n // 2) * log2n,
        "bfly_bits": 32 if ext32 else 64,
        "pass_bytes_per_launch_per_unit": None,
        "config": {"workload": f"TGGSW x TGLWE external product, N={n}, k={k}, l={l}, {total} products block-sharded over "
                               f"{world} GPU(s) by fhe_shard_range (BASELINE.json configs[3]), device-resident",
                   "n": n, "k": k, "l": l, "batch_per_gpu": batch, "global_batch": total,
                   "parallelism": f"product-sharded x{world}, no collective"},
        "global_units_per_step": total, "scaling": "strong" if args.global_batch else "weak",
    }


# Default (steps, warm-up) per config when the flags are not given.  The headline step is 23 ms: 2 + 10 steps are 0.28 s.  The
# other configurations' steps are 0.15 - 2.2 ms, and the chip takes 50 - 100 ms of load to reach its sustained clocks: with
# 2 + 10 steps config 3 reads 0.93 M ct-mul/s and config 4 1.95 M products/s where the sustained rates are 1.00 M and 2.25 M
# (same box, same process order; profiles/r05_warmup_effect.txt).  Their defaults therefore warm up for >= 0.1 s and time
# >= 0.2 s; the driver's own --steps / --warmup always win.
CONFIG_STEPS = {5: (10, 2), 2: (2000, 1000), 3: (100, 50), 4: (1000, 500)}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N ranks here — as a child
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` — BEFORE this process has imported
    torch or made any HIP call (a process that has initialised the GPU must never be replaced or forked into ranks),
    relay the children's output (rank 0 prints the one JSON line) and exit with the launcher's code."""
    import socket
    import subprocess

    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FHE_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.steps is None:
        args.steps = CONFIG_STEPS[args.config][0]
    if args.warmup is None:
        args.warmup = CONFIG_STEPS[args.config][1]
    if args.gpus < 1:
        print(f"[bench] --gpus {args.gpus}: need at least one GPU", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))                 # nothing below runs in this (GPU-free) parent
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the line's n_gpus must be what was asked for: a launcher that started a different number of ranks is an error,
        # not a note (round 4 printed a note and measured WORLD_SIZE ranks)
        if rank == 0:
            print(f"[bench] error: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                  f"start `python bench.py --gpus {args.gpus} ...` without a launcher (it starts its own ranks) or "
                  f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`", file=sys.stderr)
        sys.exit(2)
    import torch

    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective tensors live

    import fhe_study_amd as pkg  # after torch: both then share one libamdhip64

    pkg.load_library()      # raises if libfhe_ntt.so is missing: no fallback
    B = pkg.binding
    if args.batch_tile:
        B.set_batch_tile(args.batch_tile)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    if args.config == 5:
        W = workload_ntt(args, pkg, torch, dev, st, rank, world, inverse_too=False)
    elif args.config == 2:
        W = workload_ntt(args, pkg, torch, dev, st, rank, world, inverse_too=True)
    elif args.config == 3:
        W = workload_bfv(args, pkg, torch, dev, st, rank, world)
    else:
        W = workload_extprod(args, pkg, torch, dev, st, rank, world)
    step = W["step"]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Pre-warm (untimed, before the W warm-up steps): the chip needs 50 - 100 ms of load to reach its sustained clocks
    # (profiles/r05_warmup_effect.txt).  One rank's step at N = 8 is 3 ms, so W = 2 warm-up steps would start the timed
    # region on a cold chip — and a rank that finished its set-up early would have idled at the barrier below.  So: align
    # the ranks, run the step for args.prewarm seconds, then the contract's W warm-up steps, fence, K timed steps.
    prewarm_steps = 0
    if args.prewarm > 0:
        fence()
        tp = time.perf_counter()
        burst = 1
        while time.perf_counter() - tp < args.prewarm:
            for _ in range(burst):
                step()
            torch.cuda.synchronize()
            prewarm_steps += burst
            burst = min(2 * burst, 64)
    for _ in range(args.warmup):
        step()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations, measured with HIP events on the launch stream ----------
    B.kernel_timing_reset()
    B.kernel_timing_enable(True)
    prof_steps = max(1, min(3, args.steps))
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    timing = B.kernel_timing_read()
    B.kernel_timing_enable(False)
    B.kernel_timing_reset()

    out = None
    if rank == 0:
        global_units = W.get("global_units_per_step", world * W["units_per_step"])
        value = global_units * args.steps / elapsed
        per_gpu = value / world
        kernels = {k: {"avg_us": 1e3 * ms / cnt, "launches": cnt, "total_ms": ms}
                   for k, (ms, cnt) in timing.items() if cnt}
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
        step_achieved = per_gpu * W["alg_bytes_per_unit"] / 1e9      # the whole step, algorithmic GB/s per GPU
        small = W.get("bfly_bits", 64) == 32
        pm = W.get("arith") == 2
        mg = W.get("arith") == 5                                   # FHE_ARITH_MONTGOMERY (public numbering)
        valu_peak = VALU_PEAK_GBFLY32 if small else VALU_PEAK_GBFLY_PM if pm else VALU_PEAK_GBFLY_MG if mg else VALU_PEAK_GBFLY
        roofline = {
            "bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": step_achieved / HBM_PEAK_GBS, "traffic": None,
            "what": "whole step (no single streaming kernel dominates this configuration): algorithmic bytes per unit "
                    "(SURVEY.md §8d) x units per second per GPU",
            "algorithmic_bytes_per_unit": W["alg_bytes_per_unit"],
            # the whole step against the HBM roofline — for the transforms this is the number the 40 % target is about
            "step_achieved": step_achieved, "step_frac": step_achieved / HBM_PEAK_GBS,
            "dominant_kernel": dom, "kernels": kernels,
            "valu": {"bound": "integer butterflies (no MFMA on this path)", "achieved": per_gpu * W["bfly_per_unit"] / 1e9,
                     "peak": valu_peak, "unit": "Gbutterfly/s",
                     "frac": per_gpu * W["bfly_per_unit"] / 1e9 / valu_peak,
                     "butterflies_per_unit": W["bfly_per_unit"],
                     "butterfly": ("32-bit words modulo 27-bit primes (6 instructions)" if small else
                                   "64-bit words modulo a pseudo-Mersenne q = 2^k - delta (split multiplicand, 5 multiplies, 13 instructions)" if pm else
                                   "64-bit words modulo q = 1 (mod 2^32) (split multiplicand + one Montgomery word step, 5 multiplies; forward transforms)" if mg else
                                   "64-bit words modulo q < 2^61 (Shoup, 10 multiplies)"),
                     "peak_source": ("tools/ubench_bfly.hip v13 (registers only), profiles/r02_ubench_bfly.txt" if small else
                                     "tools/ubench_bfly.hip v18 (production butterfly, registers only), profiles/r04_ubench_bfly.txt" if pm else
                                     "tools/ubench_bfly.hip v14 (registers only; the kernels run its one-statement form), profiles/r04_ubench_bfly.txt" if mg else
                                     "tools/ubench_bfly.hip v8 (production butterfly, registers only), profiles/r01_ubench_bfly.txt")},
        }
        if dom and W["pass_bytes_per_launch_per_unit"]:
            # The transform configurations: `achieved` / `frac` / `traffic` describe the DOMINANT KERNEL, one launch of it
            # (algorithmic bytes per launch / its average launch duration), as the bench contract defines them.  Each
            # pass kernel of a two-pass transform reads and writes every coefficient once, so a launch is credited with
            # 16*N bytes per polynomial; the transform as a whole is `step_frac` (= transform_frac).
            launches_per_step = kernels[dom]["launches"] / prof_steps
            units_per_launch = W["batch"] / launches_per_step
            bytes_per_launch = W["pass_bytes_per_launch_per_unit"] * units_per_launch
            k_ach = bytes_per_launch / (kernels[dom]["avg_us"] * 1e-6) / 1e9
            traffic, traffic_src, step_traffic = None, None, None
            for f in PMC_TRAFFIC_FILES:
                try:
                    with open(f) as fh:
                        pmc = json.load(fh)
                    traffic = pmc["kernels"][dom]["hbm_bytes_per_polynomial"] * units_per_launch
                    # the counters come from a file an EARLIER profiling run wrote (rocprofv3 cannot run inside this process)
                    traffic_src = "prior run (" + os.path.relpath(f, ROOT) + "), not measured in this process: " + pmc["source"]
                    if all(k in pmc["kernels"] for k in kernels):   # every pass kernel touches each polynomial once per step
                        step_traffic = sum(pmc["kernels"][k]["hbm_bytes_per_polynomial"] for k in kernels) * W["batch"]
                    break
                except (OSError, KeyError, ValueError):
                    continue
            roofline.update({
                "kernel": dom, "achieved": k_ach, "frac": k_ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "what": "dominant kernel, one launch: algorithmic bytes per launch / average launch duration (HIP events); "
                        "the whole transform is step_frac",
                "avg_launch_us": kernels[dom]["avg_us"], "polys_per_launch": units_per_launch,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "step_traffic": step_traffic,      # HBM bytes of ALL the step's kernels (PMC), vs algorithmic_bytes_per_unit x units
                "transform_achieved": step_achieved, "transform_frac": step_achieved / HBM_PEAK_GBS})
        out = {
            "metric": W["metric"], "value": value, "unit": W["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm": {"seconds": args.prewarm, "steps": prewarm_steps, "what": "untimed steps before the warm-up steps (clock ramp)"},
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": W.get("scaling", "weak"), "vs_baseline": None,
            "dtype": W.get("dtype", "u64"), "data": "synthetic", "config": W["config"], "roofline": roofline,
        }

    # ---- parity against the CPU oracle (checker only; never inside the timed region) ----------
    if not args.no_parity and (rank == 0 or args.parity_all_ranks):
        import numpy as np
        from oracle import load_oracle

        rows, bad = W["parity"](load_oracle(), np)
        mine = torch.tensor([rows, bad, 1], dtype=torch.int64, device=cdev)
    else:
        mine = torch.tensor([0, 0, 0], dtype=torch.int64, device=cdev)
    if dist is not None and args.parity_all_ranks:
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
    if rank == 0 and not args.no_parity:
        rows, bad, ranks = (int(v) for v in mine.tolist())
        out["parity"] = {"rows_checked": rows, "mismatching_rows": bad, "ranks_checked": ranks,
                         "oracle": "oracle/ (CPU restatement of the reference)"}
        if bad:
            print(json.dumps(out))
            raise SystemExit(f"PARITY FAILURE: {bad} of {rows} checked units differ from the oracle")

    # ---- N > 1: shards gathered through sharding.ShardedNTT == the single-rank transform ----------
    if args.gather_check and dist is not None and args.config == 5:
        import numpy as np

        n, q, batch, plan, seed = W["n"], W["q"], W["batch"], W["plan"], W["seed"]
        gbatch = W["total"]

        def transform(rows):     # the per-rank engine: Plan.forward_dev on device tensors
            d = rows.to(dev)
            o = torch.empty_like(d)
            plan.forward_dev(d.data_ptr(), o.data_ptr(), d.shape[0], st)
            torch.cuda.synchronize()
            return o.to(cdev)

        def rows_fn(b0, b1):     # each rank generates its own rows (never broadcast)
            d = torch.empty((b1 - b0) * n, dtype=torch.int64, device=dev)
            B.fill_synthetic_dev(q, seed, b0 * n, (b1 - b0) * n, d.data_ptr(), st)
            torch.cuda.synchronize()
            return d.view(b1 - b0, n)

        eng = pkg.sharding.ShardedNTT(transform)
        full = eng.forward_sharded(rows_fn, gbatch, gather=True)
        # the same with the shard transformed a quarter at a time, every finished piece on the links while the next is transformed
        piece = max(1, -(-gbatch // world) // 4)
        over = eng.forward_sharded(rows_fn, gbatch, gather=True, overlap_rows=piece)
        if rank == 0:
            whole = transform(rows_fn(0, gbatch))
            out["gather_check"] = {"rows": gbatch, "equal_to_single_rank_transform": bool(torch.equal(full, whole)),
                                   "overlapped_pieces_of_rows": piece, "overlapped_equal": bool(torch.equal(over, whole))}

    if args.allgather and dist is not None and args.config == 5:
        # optional: the one collective of SURVEY.md §8e — every rank's REAL result shard onto every rank — timed on its
        # own (never part of `value`).  nccl = RCCL on device tensors; gloo (rehearsals) moves host copies.
        n, batch, total = W["n"], W["batch"], W["total"]
        shard = W["y"].view(batch, n)
        if args.backend != "nccl":
            shard = shard.cpu()
        chunk = args.allgather_chunk_rows
        gathered = pkg.sharding.all_gather_rows(shard, total, chunk_rows=chunk)     # warm-up: connections, buffers
        ok = None
        if rank == 0:     # the gathered batch holds this rank's rows in this rank's place (the others checked by --gather-check)
            ok = bool(torch.equal(gathered[W["b0"]:W["b0"] + batch], shard))
        del gathered
        fence()
        t0 = time.perf_counter()
        gathered = pkg.sharding.all_gather_rows(shard, total, chunk_rows=chunk)
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per = -(-total // world)
        if rank == 0:
            out["allgather"] = {"backend": "rccl" if args.backend == "nccl" else "gloo (host copies)",
                                "rows_per_rank": per, "rows_gathered": total, "collectives": 1 if not chunk else -(-per // chunk),
                                "bytes_sent_per_rank": per * n * 8, "bytes_received_per_rank": (world - 1) * per * n * 8,
                                "seconds": dt, "GBps_received_per_rank": (world - 1) * per * n * 8 / dt / 1e9,
                                "own_rows_in_place": ok,
                                "note": "timed on its own after the transform; not part of `value` (SURVEY.md §8e)"}
        del gathered

    if rank == 0 and not args.no_cpu_baseline:
        import numpy as np
        from oracle import load_oracle

        out["cpu_baseline"] = W["cpu"](load_oracle(), np, args.cpu_seconds)

    if rank == 0:
        if dist is not None:
            out["distributed"] = {"backend": "rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo",
                                  "world_size": world, "forced_at_world_1": bool(args.force_dist and world == 1)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
