#!/usr/bin/env python3
"""bench.py — NTT/s of the batched negacyclic forward NTT (N=2^16, q = 2^61-2^21+1)
on MI355X, with the kernel roofline and the CPU baseline in the same JSON line.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

A "step" is one pass of the hot path over one batch: `batch_per_gpu` independent
polynomials through fhe_ntt_forward_dev (device-resident in → out, HBM to HBM).
The batch is block-partitioned over ranks; there is no data-path collective
(SURVEY.md §8e), so per-GPU work is fixed as N grows: "scaling": "weak".
PyTorch is used for device memory, streams and torch.distributed only.

Workload = BASELINE.json configs[4] on one GPU: N=65536, batch=65536 polynomials
(32 GiB in + 32 GiB out of the 288 GB HBM), synthetic coefficients generated on
the device, outputs spot-checked against the CPU oracle.
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM traffic per launch of each kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
# WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM
# prescribes for gfx950).  Counters cannot be read from inside this process.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--q", type=int, default=Q61)
    ap.add_argument("--batch-tile", type=int, default=0, help="polynomials per launch (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--allgather", action="store_true",
                    help="also time an RCCL all-gather of a result slab (reported separately)")
    return ap.parse_args()


def cpu_baseline(q, n, seconds):
    """The oracle's reference-cost-model port (16-byte {q,v} AoS, u128 %, per-call table
    clone under a mutex — arith/src/ntt.rs:20-73) timed on this host's cores, on a bounded
    sample of the same synthetic workload."""
    from oracle import load_oracle

    O = load_oracle()
    O.roots(q, n)  # table build is a one-off in the reference too (CACHE), keep it out of the timing
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)   # the GPU box's CPU share for one GPU
    probe = O.fill_synthetic(q, 0xF4E50005, 0, 4 * n)
    t0 = time.perf_counter()
    O.ref_ntt_aos(q, n, probe, threads=1)
    per_ntt = (time.perf_counter() - t0) / 4
    one_core = 1.0 / per_ntt
    count = max(cores, int(seconds / per_ntt) // cores * cores)
    count = min(count, 8192)
    x = O.fill_synthetic(q, 0xF4E50005, 0, count * n)
    t0 = time.perf_counter()
    O.ref_ntt_aos(q, n, x, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": count / dt, "unit": "NTT/s", "cores": cores, "kind": "port",
        "sample": f"{count} forward NTTs of the same synthetic rows (N={n}), {cores} threads, {dt:.1f} s; "
                  f"single thread: {one_core:.1f} NTT/s",
        "value_1core": one_core,
    }


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
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

    lib = pkg.load_library()      # raises if libfhe_ntt.so is missing: no fallback
    B = pkg.binding
    q, n = args.q, 1 << args.log_n
    batch = args.batch_per_gpu
    plan = pkg.Plan(q, n)
    if args.batch_tile:
        B.set_batch_tile(args.batch_tile)

    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    x = torch.empty(batch * n, dtype=torch.int64, device=dev)
    y = torch.empty(batch * n, dtype=torch.int64, device=dev)
    seed = 0xF4E50005
    first = rank * batch * n      # rank r owns rows [r*batch, (r+1)*batch) of the global batch
    B.fill_synthetic_dev(q, seed, first, batch * n, x.data_ptr(), st)

    def step():
        plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

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
        total_ntts = world * batch * args.steps
        value = total_ntts / elapsed
        alg_bytes_per_ntt = 16 * n   # read N + write N coefficients of 8 B
        kernels = {k: {"avg_us": 1e3 * ms / cnt, "launches": cnt, "total_ms": ms}
                   for k, (ms, cnt) in timing.items() if cnt}
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
        roofline = None
        if dom:
            launches_per_step = kernels[dom]["launches"] / prof_steps
            polys_per_launch = batch / launches_per_step
            # each pass kernel reads and writes every coefficient of its polynomials once
            bytes_per_launch = alg_bytes_per_ntt * polys_per_launch
            achieved = bytes_per_launch / (kernels[dom]["avg_us"] * 1e-6) / 1e9
            traffic, traffic_src = None, None
            try:
                with open(PMC_TRAFFIC_FILE) as f:
                    pmc = json.load(f)
                per_poly = pmc["kernels"][dom]["hbm_bytes_per_polynomial"]
                traffic = per_poly * polys_per_launch
                traffic_src = pmc["source"]
            except (OSError, KeyError, ValueError):
                pass
            roofline = {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_us": kernels[dom]["avg_us"], "polys_per_launch": polys_per_launch,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # the whole transform (all its kernels): 16*N bytes per NTT over the timed region
                "transform_achieved": (value / world) * alg_bytes_per_ntt / 1e9,
                "transform_frac": (value / world) * alg_bytes_per_ntt / 1e9 / HBM_PEAK_GBS,
                "kernels": kernels,
            }
        out = {
            "metric": "NTT/s (N=2^16, 64-bit q) per node; achieved HBM GB/s vs roofline",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"batched forward negacyclic NTT, N={n}, q={q}, "
                                   f"{batch} polynomials per GPU (BASELINE.json configs[4] shape), "
                                   "device-resident in->out",
                       "n": n, "q": q, "batch_per_gpu": batch, "global_batch": world * batch,
                       "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": roofline,
        }

    # ---- parity subset: first 8, last 8, 48 pseudo-random rows ----------
    if rank == 0 and not args.no_parity:
        import numpy as np
        from oracle import load_oracle

        O = load_oracle()
        rng = np.random.default_rng(7)
        rows = sorted(set(list(range(min(8, batch))) + list(range(max(0, batch - 8), batch)) +
                          [int(r) for r in rng.integers(0, batch, 48)]))
        Y = y.view(batch, n)
        bad = 0
        for r in rows:
            a = O.fill_synthetic(q, seed, first + r * n, n)
            if not np.array_equal(Y[r].cpu().numpy().view(np.uint64), O.ntt(q, n, a)):
                bad += 1
        out["parity"] = {"rows_checked": len(rows), "mismatching_rows": bad, "oracle": "oracle/ntt_oracle.c"}
        if bad:
            print(json.dumps(out))
            raise SystemExit(f"PARITY FAILURE: {bad} of {len(rows)} rows differ from the oracle")

    if args.allgather and dist is not None and args.backend == "nccl":
        # optional: the one collective of SURVEY.md §8e, timed on its own (not part of `value`)
        slab = min(batch, 1024) * n
        gathered = torch.empty(world * slab, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, y[:slab])
        fence()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, y[:slab])
        fence()
        dt = time.perf_counter() - t0
        if rank == 0:
            out["allgather"] = {"bytes_received_per_rank": (world - 1) * slab * 8, "seconds": dt,
                                "GBps_per_rank": (world - 1) * slab * 8 / dt / 1e9}

    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(q, n, args.cpu_seconds)

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
