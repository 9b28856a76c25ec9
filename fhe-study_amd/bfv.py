"""Host mirror of the reference's BFV ciphertext multiply surface (bfv/src/lib.rs) over the
C ABI.  Only the product path lives here — `RLWE::tensor`, `BFV::relinearize_204`,
`RLWE::mul` (lib.rs:59-90, 251-271); key generation / encryption stay with the reference.

    reference (Rust)                         here
    RLWE(Rq, Rq)                lib.rs:47    RLWE(c0, c1)          (coefficients mod q)
    RLK(Rq, Rq)  mod p*q        lib.rs:43    RLK(rlk0, rlk1, pq)
    RLWE::tensor(t, &a, &b)     lib.rs:59    RLWE.tensor(t, a, b) -> (c0, c1, c2)
    RLWE::mul(t, &rlk, &a, &b)  lib.rs:87    RLWE.mul(t, rlk, a, b)
"""
from dataclasses import dataclass

import numpy as np

from . import binding
from .arith import RingParam, Rq


@dataclass
class RLK:
    """relinearisation key, coefficients mod p*q (lib.rs:41-43)"""
    rlk0: np.ndarray
    rlk1: np.ndarray
    pq: int


class RLWE:
    """RLWE ciphertext (c0, c1), lib.rs:45-47; `c0`/`c1` are arith.Rq (or batches of them)."""

    def __init__(self, c0, c1):
        if c0.param != c1.param:
            raise binding.FheError(binding.FHE_E_PARAM_MISMATCH, "RLWE components differ in RingParam")
        self.c0, self.c1 = c0, c1

    @property
    def param(self):
        return self.c0.param

    @staticmethod
    def tensor(t, a, b):
        """lib.rs:59-85 → (c0, c1, c2) as Rq mod q"""
        p = a.param
        if b.param != p:
            raise binding.FheError(binding.FHE_E_PARAM_MISMATCH, "operands differ in RingParam")
        c = binding.bfv_tensor(p.q, p.n, t, a.c0.coeffs, a.c1.coeffs, b.c0.coeffs, b.c1.coeffs)
        shape = a.c0.coeffs.shape
        return tuple(Rq(p, x.reshape(shape)) for x in c)

    @staticmethod
    def mul(t, rlk, a, b):
        """lib.rs:87-90: relinearize_204(rlk, tensor(t, a, b))"""
        p = a.param
        if b.param != p:
            raise binding.FheError(binding.FHE_E_PARAM_MISMATCH, "operands differ in RingParam")
        o0, o1 = binding.bfv_mul(p.q, p.n, t, rlk.pq, rlk.rlk0, rlk.rlk1,
                                 a.c0.coeffs, a.c1.coeffs, b.c0.coeffs, b.c1.coeffs)
        shape = a.c0.coeffs.shape
        return RLWE(Rq(p, o0.reshape(shape)), Rq(p, o1.reshape(shape)))
