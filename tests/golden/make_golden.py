#!/usr/bin/env python3
"""Generates tests/golden/*.npz and tests/golden/kat.json.

The reference is Rust and cannot be built or imported in this image (no
cargo/rustc), so these vectors come from the pure-Python restatement
oracle/ntt_oracle_py.py (arbitrary-precision ints, line-by-line twin of
arith/src/ntt.rs, zq.rs, ring_nq.rs) — NOT from the C oracle, so that the C
oracle, the plan builder and the HIP kernels are all checked against a second,
independent implementation.  The reference's literal known-answer vectors
(arith/src/ring_nq.rs:674-682) and the Sage input pair
(arith/sage/ring.sage:20-22) are stored verbatim in kat.json.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ntt_oracle_py as P  # noqa: E402

Q16 = 65537
Q61 = 2305843009211596801  # 2^61 - 2^21 + 1


def u64(x):
    return np.asarray(x, dtype=np.uint64)


def case(q, n, seed, rows):
    rts, rts_inv, n_inv, psi = P.roots(q, n)
    a = [P.fill_synthetic(q, seed, r * n, n) for r in range(rows)]
    b = [P.fill_synthetic(q, seed ^ 0xB0B, r * n, n) for r in range(rows)]
    A = [P.ntt(q, n, rts, x) for x in a]
    B = [P.ntt(q, n, rts, x) for x in b]
    back = [P.intt(q, n, rts_inv, n_inv, x) for x in A]
    assert back == a
    C = [[(x * y) % q for x, y in zip(ra, rb)] for ra, rb in zip(A, B)]
    c = [P.intt(q, n, rts_inv, n_inv, x) for x in C]
    if n <= 512:
        for x, y, z in zip(a, b, c):
            assert P.naive_negacyclic_mul(q, n, x, y) == z
    return dict(q=u64(q), n=u64(n), psi=u64(psi), n_inv=u64(n_inv), seed=u64(seed),
                roots=u64(rts), roots_inv=u64(rts_inv), a=u64(a), b=u64(b),
                ntt_a=u64(A), ntt_b=u64(B), c=u64(c), c_evals=u64(C))


def main():
    cases = {
        "q16_n4": (Q16, 4, 0xF4E50001, 4),
        "q16_n8": (Q16, 8, 0xF4E50002, 4),
        "q16_n512": (Q16, 512, 0xF4E50003, 3),
        "q61_n2": (Q61, 2, 0xF4E50004, 4),
        "q61_n64": (Q61, 64, 0xF4E50005, 3),
        "q61_n1024": (Q61, 1024, 0xF4E50006, 2),
        "q61_n4096": (Q61, 4096, 0xF4E50007, 2),
    }
    for name, (q, n, seed, rows) in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **case(q, n, seed, rows))
        print("wrote", name)

    # Large sizes: digests only (tables are MiB-sized).
    digests = {}
    for q, n, seed in ((Q61, 8192, 0xF4E50008), (Q61, 65536, 0xF4E50009), (Q16, 32768, 0xF4E5000A)):
        rts, rts_inv, n_inv, psi = P.roots(q, n)
        a = P.fill_synthetic(q, seed, 0, n)
        A = P.ntt(q, n, rts, a)
        digests[f"q{q}_n{n}"] = dict(
            q=q, n=n, seed=seed, psi=psi, n_inv=n_inv,
            roots_sha256=hashlib.sha256(u64(rts).tobytes()).hexdigest(),
            roots_inv_sha256=hashlib.sha256(u64(rts_inv).tobytes()).hexdigest(),
            a_sha256=hashlib.sha256(u64(a).tobytes()).hexdigest(),
            ntt_a_sha256=hashlib.sha256(u64(A).tobytes()).hexdigest(),
            ntt_a_head=[int(x) for x in A[:8]],
        )
        print("digest", q, n)

    kat = {
        "_source": "literal vectors held by the reference's own tests",
        "ring_nq_test_mul": [  # arith/src/ring_nq.rs:674-682, q = 2^16+1, n = 4, via mul_mut
            {"q": Q16, "n": 4, "a": [1, 2, 3, 4], "b": [1, 2, 3, 4], "c": [65513, 65517, 65531, 20]},
            {"q": Q16, "n": 4, "a": [0, 0, 0, 2], "b": [0, 0, 0, 2], "c": [0, 0, 65533, 0]},
        ],
        "ntt_test_ntt_roundtrip": {"q": Q16, "n": 4, "a": [1, 2, 3, 4]},  # arith/src/ntt.rs:194-215
        "ntt_test_ntt_loop": {"q": Q16, "n": 512, "iters": 1000},  # arith/src/ntt.rs:217-234
        "sage_ring_inputs": {"q": Q16, "n": 4, "a": [4, 2, 1, 0], "b": [1, 2, 3, 4]},  # arith/sage/ring.sage:20-22 (output not recorded in-tree)
        "ring_n_test_mul_note": "arith/src/ring_n.rs:453-470 pins the Z schoolbook (next-row N1)",
        "digests": digests,
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote kat.json")


if __name__ == "__main__":
    main()
