"""tools/_timing.py — one timing rule for the diagnostic tools: the chip needs 50 - 100 ms of load to reach its sustained
clocks (profiles/r05_warmup_effect.txt: a 2-call warm-up understated short kernels by 5 - 15 %), so a measurement warms up
for at least `warm_s` seconds of back-to-back calls and then times whole calls for at least `time_s` seconds."""
import time

import torch


def timeit(f, warm_s=0.2, time_s=0.3, min_reps=3):
    """seconds per call of f() (device work enqueued on any stream; synchronised here)"""
    t0 = time.perf_counter()
    n = 0
    while True:                                    # warm-up by wall time, in growing bursts
        for _ in range(max(1, n)): f()
        torch.cuda.synchronize()
        n = max(1, 2 * n)
        if time.perf_counter() - t0 >= warm_s:
            break
    f(); torch.cuda.synchronize()
    t1 = time.perf_counter(); f(); torch.cuda.synchronize()
    one = max(time.perf_counter() - t1, 1e-6)
    reps = max(min_reps, int(time_s / one))
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
