"""Host-side mirror of the reference's `arith::{RingParam, Rq, NTT}` surface for the
hot path, over the C ABI (binding.py → libfhe_ntt.so → HIP kernels).

Same names, argument meaning and error behaviour as the reference so that the
parity tests read like the reference's own tests:

    reference (Rust)                                   here
    ------------------------------------------------   --------------------------
    RingParam{q,n}            arith/src/ring.rs:6-10   RingParam(q, n)
    Rq::from_vec_u64          ring_nq.rs:160-163       Rq.from_vec_u64(param, coeffs)
    Rq::compute_evals         ring_nq.rs:147-150       Rq.compute_evals()
    NTT::ntt / NTT::intt      ntt.rs:44,78             NTT.ntt(a) / NTT.intt(a)
    &a * &b, mul              ring_nq.rs:490-503,586   a * b, mul(a, b)
    a.mul(&mut b), mul_mut    ring_nq.rs:294-296,564   a.mul(b), mul_mut(a, b)
    panic!/assert!            ntt.rs:116-130, ring_nq.rs:565,587   raises binding.FheError

Only the NTT path lives here; the O(N) glue of ring_nq.rs (add, sub, decompose,
mod_switch, ...) is out of scope and stays with the reference.
"""
from dataclasses import dataclass

import numpy as np

from . import binding
from .binding import FheError, Plan

_plans = {}


def _plan(param):
    key = (param.q, param.n)
    if key not in _plans:
        _plans[key] = Plan(param.q, param.n)
    return _plans[key]


@dataclass(frozen=True)
class RingParam:
    """arith/src/ring.rs:6-10"""
    q: int
    n: int


class Rq:
    """Element of Z_q[X]/(X^n+1): `coeffs` (canonical u64 values) + optional cached
    `evals` (ring_nq.rs:19-27).  Also usable as a batch: coeffs of shape (b, n)."""

    __slots__ = ("param", "coeffs", "evals")

    def __init__(self, param, coeffs, evals=None):
        self.param = param
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        self.evals = None if evals is None else np.ascontiguousarray(evals, dtype=np.uint64)
        if self.coeffs.shape[-1] != param.n:
            raise ValueError("coefficient vector length != n")

    @staticmethod
    def from_vec_u64(param, coeffs):
        """ring_nq.rs:160-163: values are reduced mod q (Zq::from_u64, zq.rs:21-31);
        the X^n+1 fold of ring_nq.rs:132-141 applies to vectors longer than n."""
        c = [int(x) % param.q for x in coeffs]
        n = param.n
        if len(c) > n:
            for i in range(n, len(c)):
                c[i - n] = (c[i - n] - c[i]) % param.q
            c = c[:n]
        elif len(c) < n:
            raise ValueError("fewer than n coefficients")
        return Rq(param, np.array(c, dtype=np.uint64))

    def compute_evals(self):
        """ring_nq.rs:147-150"""
        self.evals = NTT.ntt(self).coeffs

    def mul(self, rhs):
        """Rq::mul(&mut self, &mut rhs), ring_nq.rs:294-296"""
        return mul_mut(self, rhs)

    def __mul__(self, rhs):
        return mul(self, rhs)

    def __eq__(self, other):
        """ring_nq.rs:401-405: coefficients and param; evals are not compared"""
        return (isinstance(other, Rq) and self.param == other.param
                and np.array_equal(self.coeffs, other.coeffs))

    def __repr__(self):
        return f"Rq(q={self.param.q}, n={self.param.n}, coeffs={self.coeffs!r})"


class NTT:
    """arith/src/ntt.rs:14"""

    @staticmethod
    def ntt(a):
        """ntt.rs:44-73 → Rq whose `coeffs` are the NTT-domain values, evals=None"""
        return Rq(a.param, _plan(a.param).forward(a.coeffs).reshape(a.coeffs.shape), None)

    @staticmethod
    def intt(a):
        """ntt.rs:78-110"""
        return Rq(a.param, _plan(a.param).inverse(a.coeffs).reshape(a.coeffs.shape), None)


def _check_param(lhs, rhs):
    if lhs.param != rhs.param:
        # assert_eq!(lhs.param, rhs.param), ring_nq.rs:565,587
        raise FheError(binding.FHE_E_PARAM_MISMATCH,
                       f"operands have different RingParam: {lhs.param} vs {rhs.param}")


def mul(lhs, rhs):
    """ring_nq.rs:586-607: uses cached evals where present; the product carries its evals."""
    _check_param(lhs, rhs)
    a, a_ev = (lhs.evals, True) if lhs.evals is not None else (lhs.coeffs, False)
    b, b_ev = (rhs.evals, True) if rhs.evals is not None else (rhs.coeffs, False)
    c, c_evals, _, _ = _plan(lhs.param).rq_mul(a, b, a_ev, b_ev)
    shape = lhs.coeffs.shape
    return Rq(lhs.param, c.reshape(shape), c_evals.reshape(shape))


def mul_mut(lhs, rhs):
    """ring_nq.rs:564-583: like `mul`, and stores the operands' evals back into them."""
    _check_param(lhs, rhs)
    a, a_ev = (lhs.evals, True) if lhs.evals is not None else (lhs.coeffs, False)
    b, b_ev = (rhs.evals, True) if rhs.evals is not None else (rhs.coeffs, False)
    c, c_evals, a_evals, b_evals = _plan(lhs.param).rq_mul(a, b, a_ev, b_ev)
    shape = lhs.coeffs.shape
    if lhs.evals is None:
        lhs.evals = a_evals.reshape(shape)
    if rhs.evals is None:
        rhs.evals = b_evals.reshape(shape)
    return Rq(lhs.param, c.reshape(shape), c_evals.reshape(shape))


def pm_params(q):
    """The pseudo-Mersenne form the engine detects at plan time (csrc/capi.hip: build_plan, csrc/zq_device.hpp):
    q = 2^k - delta with 56 <= k <= 61 and delta <= 2^(k-39).  Such a modulus (2^61 - 2^21 + 1 is one) runs the
    five-multiply butterflies on tables {w, w 2^32 mod q}; returns the kernels' constants, or None.  Host-side
    restatement for the tests and for callers choosing a modulus; the library decides on its own."""
    k = int(q).bit_length()
    delta = (1 << k) - q
    if not (56 <= k <= 61 and delta <= 1 << (k - 39)):
        return None
    return {"k": k, "delta": delta, "c2": 2 * delta, "sh": k - 31, "mask": (1 << (k - 31)) - 1,
            "rsh": k - 32, "rmask": (1 << (k - 32)) - 1}


def mg_params(q):
    """The word-Montgomery form the engine detects at plan time (csrc/capi.hip: build_plan, csrc/zq_device.hpp: ct_bfly_mg):
    q = qh 2^32 + 1 below 2^61 (and not pseudo-Mersenne, which cannot coincide).  Such a modulus runs its transforms (both directions) and Rq products
    on tables {w 2^32 mod q, w 2^64 mod q}: q^-1 = 1 (mod 2^32), so a Montgomery word step needs no multiplication by it.
    Returns the kernels' constant (2^32 - qh), or None.  Host-side restatement for the tests; the library decides on its own."""
    q = int(q)
    if (q & 0xffffffff) != 1 or (q >> 32) == 0 or (q >> 61) != 0 or pm_params(q) is not None:
        return None
    return {"qh": q >> 32, "nqh": (1 << 32) - (q >> 32)}
