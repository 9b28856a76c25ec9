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
        _i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_r_naive_mul.restype = None
        L.oracle_r_naive_mul.argtypes = [_u64, _i64p, _i64p, _i64p]
        L.oracle_r_modulus.restype = None
        L.oracle_r_modulus.argtypes = [_u64, _i64p, _u64, _i64p]
        L.oracle_mul_div_round.restype = None
        L.oracle_mul_div_round.argtypes = [_u64, _u64, _i64p, _u64, _u64, _u64, _p64]
        L.oracle_zq_from_f64.restype = _u64
        L.oracle_zq_from_f64.argtypes = [_u64, ctypes.c_double]
        L.oracle_bfv_tensor.restype = None
        L.oracle_bfv_tensor.argtypes = [_u64, _u64, _u64] + [_p64] * 7
        L.oracle_bfv_relinearize_204.restype = None
        L.oracle_bfv_relinearize_204.argtypes = [_u64, _u64, _u64] + [_p64] * 7
        L.oracle_bfv_mul.restype = None
        L.oracle_bfv_mul.argtypes = [_u64, _u64, _u64, _u64] + [_p64] * 8
        L.oracle_tn_mul.restype = None
        L.oracle_tn_mul.argtypes = [_u64, _p64, _p64, _p64]
        L.oracle_tn_decompose.restype = None
        L.oracle_tn_decompose.argtypes = [_u64, ctypes.c_uint32, _p64, _p64]
        L.oracle_external_product.restype = None
        L.oracle_external_product.argtypes = [_u64, ctypes.c_uint32, ctypes.c_uint32, _p64, _p64, _p64]
        for name, at in (("oracle_rq_add", [_u64, _u64, _p64, _p64, _p64]), ("oracle_rq_sub", [_u64, _u64, _p64, _p64, _p64]),
                         ("oracle_rq_neg", [_u64, _u64, _p64, _p64]), ("oracle_rq_mul_by_u64", [_u64, _u64, _p64, _u64, _p64]),
                         ("oracle_rq_mod_switch", [_u64, _u64, _p64, _u64, _p64]),
                         ("oracle_rq_mul_div_round", [_u64, _u64, _p64, _u64, _u64, _p64]),
                         ("oracle_rq_decompose", [_u64, _u64, _p64, ctypes.c_uint32, ctypes.c_uint32, _p64]),
                         ("oracle_rq_remodule", [_u64, _p64, _u64, _p64]),
                         ("oracle_rq_mul_by_f64", [_u64, _u64, _p64, ctypes.c_double, _p64]),
                         ("oracle_rq_div_round", [_u64, _u64, _p64, _u64, _p64])):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = at
        u32 = ctypes.c_uint32
        for name, at in (("oracle_tr_dot", [_u64, _u64, u32, _p64, _p64, _p64]), ("oracle_tr_mul_r", [_u64, _u64, u32, _p64, _p64, _p64]),
                         ("oracle_glev_mul", [_u64, _u64, u32, u32, _p64, _p64, _p64]),
                         ("oracle_key_switch", [_u64, _u64, u32, u32, u32, _p64, _p64, _p64])):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = at
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

    # -- next rows (oracle/fhe_next_oracle.c) -----------------------------------
    def r_naive_mul(self, n, a, b):
        a = np.ascontiguousarray(a, dtype=np.int64).reshape(-1, n)
        b = np.ascontiguousarray(b, dtype=np.int64).reshape(-1, n)
        out = np.empty((a.shape[0], 2 * n - 1), dtype=np.int64)
        i64 = ctypes.POINTER(ctypes.c_int64)
        for i in range(a.shape[0]):
            self.lib.oracle_r_naive_mul(n, a[i].ctypes.data_as(i64), b[i].ctypes.data_as(i64),
                                        out[i].ctypes.data_as(i64))
        return out

    def r_modulus(self, n, p):
        p = np.ascontiguousarray(p, dtype=np.int64)
        out = np.empty(n, dtype=np.int64)
        i64 = ctypes.POINTER(ctypes.c_int64)
        self.lib.oracle_r_modulus(n, p.ctypes.data_as(i64), p.size, out.ctypes.data_as(i64))
        return out

    def mul_div_round(self, q, n, v, num, den):
        v = np.ascontiguousarray(v, dtype=np.int64).reshape(-1, 2 * n - 1)
        out = np.empty((v.shape[0], n), dtype=np.uint64)
        i64 = ctypes.POINTER(ctypes.c_int64)
        for i in range(v.shape[0]):
            self.lib.oracle_mul_div_round(q, n, v[i].ctypes.data_as(i64), 2 * n - 1, num, den, _ptr(out[i]))
        return out

    def bfv_tensor(self, q, n, t, a0, a1, b0, b1):
        A = [_arr(x).reshape(-1, n) for x in (a0, a1, b0, b1)]
        batch = A[0].shape[0]
        c = np.empty((3, batch, n), dtype=np.uint64)
        for i in range(batch):
            self.lib.oracle_bfv_tensor(q, n, t, _ptr(A[0][i]), _ptr(A[1][i]), _ptr(A[2][i]), _ptr(A[3][i]),
                                       _ptr(c[0][i]), _ptr(c[1][i]), _ptr(c[2][i]))
        return c[0], c[1], c[2]

    def bfv_mul(self, q, n, t, pq, rlk0, rlk1, a0, a1, b0, b1):
        A = [_arr(x).reshape(-1, n) for x in (a0, a1, b0, b1)]
        r0, r1 = _arr(rlk0), _arr(rlk1)
        batch = A[0].shape[0]
        o = np.empty((2, batch, n), dtype=np.uint64)
        for i in range(batch):
            self.lib.oracle_bfv_mul(q, n, t, pq, _ptr(r0), _ptr(r1), _ptr(A[0][i]), _ptr(A[1][i]),
                                    _ptr(A[2][i]), _ptr(A[3][i]), _ptr(o[0][i]), _ptr(o[1][i]))
        return o[0], o[1]

    def tn_mul(self, n, a, b):
        a = _arr(a).reshape(-1, n)
        b = _arr(b).reshape(-1, n)
        out = np.empty_like(a)
        for i in range(a.shape[0]):
            self.lib.oracle_tn_mul(n, _ptr(a[i]), _ptr(b[i]), _ptr(out[i]))
        return out

    def external_product(self, n, k, l, tggsw, tglwe):
        g = _arr(tggsw)
        t = _arr(tglwe).reshape(-1, (k + 1) * n)
        out = np.empty_like(t)
        for i in range(t.shape[0]):
            self.lib.oracle_external_product(n, k, l, _ptr(g), _ptr(t[i]), _ptr(out[i]))
        return out.reshape(-1, k + 1, n)

    # -- rows N3/N4 (oracle/fhe_glue_oracle.c) -------------------------------------
    def glue(self, name, *args):
        """call oracle_<name>(...) with numpy arrays converted to u64 pointers; returns nothing"""
        conv = [(_ptr(a) if isinstance(a, np.ndarray) else a) for a in args]
        rc = getattr(self.lib, "oracle_" + name)(*conv)
        if rc not in (None, 0):
            raise ValueError(f"oracle_{name} failed")

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
