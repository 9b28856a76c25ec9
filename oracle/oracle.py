"""ctypes front-end for oracle/liboracle.so (oracle/ntt_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of oracle/ntt_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

_u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(ctypes.c_uint64)


def build_oracle(force=False):
    """Compile oracle/*.c with gcc (make).  Returns the library path."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(_LIB)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(_p64)


def _arr(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    if shape is not None:
        a = a.reshape(shape)
    return a


class Oracle:
    """Thin, stateless wrapper; tables are recomputed on demand and memoised here
    (the memo mirrors the reference's CACHE, arith/src/ntt.rs:18)."""

    def __init__(self, lib_path=None):
        self.lib = ctypes.CDLL(lib_path or build_oracle())
        L = self.lib
        L.oracle_exp_mod.restype = _u64
        L.oracle_exp_mod.argtypes = [_u64, _u64, _u64]
        L.oracle_inv_mod.restype = _u64
        L.oracle_inv_mod.argtypes = [_u64, _u64]
        L.oracle_zq_mul.restype = _u64
        L.oracle_zq_mul.argtypes = [_u64, _u64, _u64]
        L.oracle_primitive_root_of_unity.restype = _u64
        L.oracle_primitive_root_of_unity.argtypes = [_u64, _u64]
        L.oracle_roots.restype = ctypes.c_int
        L.oracle_roots.argtypes = [_u64, _u64, _p64, _p64, _p64, _p64]
        L.oracle_ntt_batch.restype = None
        L.oracle_ntt_batch.argtypes = [_u64, _u64, _p64, _p64, _p64, _u64]
        L.oracle_intt_batch.restype = None
        L.oracle_intt_batch.argtypes = [_u64, _u64, _p64, _u64, _p64, _p64, _u64]
        L.oracle_rq_mul.restype = None
        L.oracle_rq_mul.argtypes = [_u64, _u64, _p64, _p64, _u64] + [_p64] * 6
        L.oracle_pointwise_mul.restype = None
        L.oracle_pointwise_mul.argtypes = [_u64, _u64, _p64, _p64, _p64]
        L.oracle_naive_negacyclic_mul.restype = None
        L.oracle_naive_negacyclic_mul.argtypes = [_u64, _u64, _p64, _p64, _p64]
        L.oracle_ref_ntt_batch_aos.restype = ctypes.c_int
        L.oracle_ref_ntt_batch_aos.argtypes = [_u64, _u64, _p64, _p64, _p64, _p64, _u64,
                                               ctypes.c_int]
        L.oracle_fill_synthetic.restype = None
        L.oracle_fill_synthetic.argtypes = [_u64, _u64, _u64, _u64, _p64]
        self._tables = {}

    # -- tables ---------------------------------------------------------------
    def roots(self, q, n):
        """(roots, roots_inv, n_inv, psi) — raises ValueError where the reference panics."""
        key = (int(q), int(n))
        if key not in self._tables:
            r = np.zeros(max(n, 1), dtype=np.uint64)
            ri = np.zeros(max(n, 1), dtype=np.uint64)
            n_inv = _u64(0)
            psi = _u64(0)
            rc = self.lib.oracle_roots(q, n, _ptr(r), _ptr(ri), ctypes.byref(n_inv),
                                       ctypes.byref(psi))
            if rc != 0:
                raise ValueError(f"reference would panic for (q={q}, n={n})")
            self._tables[key] = (r, ri, int(n_inv.value), int(psi.value))
        return self._tables[key]

    # -- transforms -----------------------------------------------------------
    def ntt(self, q, n, a):
        a = _arr(a)
        batch = a.size // n
        r, _, _, _ = self.roots(q, n)
        out = np.empty_like(a)
        self.lib.oracle_ntt_batch(q, n, _ptr(r), _ptr(a), _ptr(out), batch)
        return out

    def intt(self, q, n, a):
        a = _arr(a)
        batch = a.size // n
        _, ri, n_inv, _ = self.roots(q, n)
        out = np.empty_like(a)
        self.lib.oracle_intt_batch(q, n, _ptr(ri), n_inv, _ptr(a), _ptr(out), batch)
        return out

    def rq_mul(self, q, n, a, b):
        """batched Rq multiply → (c, c_evals, a_evals, b_evals)"""
        a = _arr(a)
        b = _arr(b)
        assert a.shape == b.shape
        batch = a.size // n
        r, ri, n_inv, _ = self.roots(q, n)
        c = np.empty_like(a)
        ce = np.empty_like(a)
        ae = np.empty_like(a)
        be = np.empty_like(a)
        af, bf, cf, cef, aef, bef = (x.reshape(batch, n) for x in (a, b, c, ce, ae, be))
        for i in range(batch):
            self.lib.oracle_rq_mul(q, n, _ptr(r), _ptr(ri), n_inv, _ptr(af[i]), _ptr(bf[i]),
                                   _ptr(cf[i]), _ptr(cef[i]), _ptr(aef[i]), _ptr(bef[i]))
        return c, ce, ae, be

    def pointwise_mul(self, q, a, b):
        a = _arr(a)
        b = _arr(b)
        c = np.empty_like(a)
        self.lib.oracle_pointwise_mul(q, a.size, _ptr(a), _ptr(b), _ptr(c))
        return c

    def naive_negacyclic_mul(self, q, n, a, b):
        a = _arr(a)
        b = _arr(b)
        batch = a.size // n
        c = np.empty_like(a)
        af, bf, cf = (x.reshape(batch, n) for x in (a, b, c))
        for i in range(batch):
            self.lib.oracle_naive_negacyclic_mul(q, n, _ptr(af[i]), _ptr(bf[i]), _ptr(cf[i]))
        return c

    def ref_ntt_aos(self, q, n, a, threads=1):
        """Forward NTT with the reference's cost model (CPU baseline leg)."""
        a = _arr(a)
        batch = a.size // n
        r, ri, _, _ = self.roots(q, n)
        out = np.empty_like(a)
        self.lib.oracle_ref_ntt_batch_aos(q, n, _ptr(r), _ptr(ri), _ptr(a), _ptr(out), batch,
                                          int(threads))
        return out

    def fill_synthetic(self, q, seed, first_index, count):
        out = np.empty(count, dtype=np.uint64)
        self.lib.oracle_fill_synthetic(q, seed, first_index, count, _ptr(out))
        return out

    def exp_mod(self, q, x, k):
        return int(self.lib.oracle_exp_mod(q, x, k))

    def inv_mod(self, q, x):
        return int(self.lib.oracle_inv_mod(q, x))


_singleton = None


def load_oracle():
    global _singleton
    if _singleton is None:
        _singleton = Oracle()
    return _singleton
