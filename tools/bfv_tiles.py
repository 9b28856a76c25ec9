#!/usr/bin/env python3
"""tools/bfv_tiles.py — experiment (round 5, NEGATIVE): 2048 BFV ciphertext products (N=8192, q=65537, p=q^2) as tiles of T
pairs dealt round-robin over S streams, so that a tile's intermediates (512 KiB of transforms per pair) stay in the 256 MB
Infinity Cache between its four kernels while another stream's kernels fill the tails.  Prints ms per 2048 pairs per (T, S).
The first version of this script timed 5 calls after 2 warm-up calls and showed +10 % for tiles of 256 on 3 streams — the
first configuration in the list had simply run on a chip that had not reached its clocks.  With tools/_timing.py (and with
the tiling built into the library and each setting in its own process: gpurun_out/r5r) tiles are 1.5 - 6 % SLOWER than one
launch per kernel: profiles/r05_bfv_tiles_ab.txt.  The library does not tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fhe_study_amd as pkg
from _timing import timeit
B, L = pkg.binding, pkg.load_library()
n, q, t, batch = 8192, 65537, 2, 2048
pq = q * q * q
rng = np.random.default_rng(5)
rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).cuda()
full = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).cuda()
for T, S in [(2048, 1), (1024, 2), (512, 2), (256, 2), (256, 3), (192, 2), (128, 2), (128, 3), (128, 4), (96, 3), (64, 4), (256, 1), (128, 1)]:
    tiles = [full[:, i:i + T, :].contiguous() for i in range(0, batch, T)]
    outs = [torch.empty((2, x.shape[1], n), dtype=torch.int64, device="cuda") for x in tiles]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run():
        for i, (x, o) in enumerate(zip(tiles, outs)):
            st = streams[i % S]
            B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), x.data_ptr(), o.data_ptr(), x.shape[1], st.cuda_stream))
    dt = timeit(run)
    print(f"tile {T:5d} pairs x {S} streams: {dt*1e3:7.3f} ms per {batch} pairs  {batch/dt/1e6:.3f} M ct-mul/s", flush=True)
    del tiles, outs
