"""Host mirror of the reference's torus products (arith/src/ring_torus.rs, tfhe/src/tggsw.rs)
over the C ABI.

    reference (Rust)                                   here
    Tn * Tn           ring_torus.rs:251-298            Tn.__mul__
    Tn::decompose     ring_torus.rs:67-77              (inside the external product, on the GPU)
    TGLWE(GLWE<Tn>)   tfhe/src/tglwe.rs:33             TGLWE(a [k][n], b [n])
    TGLWE * Tn        tfhe/src/tglwe.rs:182-194        TGLWE.__mul__
    TGLev * Vec<Tn>   tfhe/src/tggsw.rs:139-149        TGLev.__mul__
    TGGSW * TGLWE     tfhe/src/tggsw.rs:45-62          TGGSW.__mul__   (beta = 2, l = 64 as there)
"""
import numpy as np

from . import binding


class Tn:
    """element of T_{2^64}[X]/(X^n+1): u64 coefficients, arithmetic wraps (torus.rs:80-153)"""

    def __init__(self, coeffs):
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)

    @property
    def n(self):
        return self.coeffs.shape[-1]

    def __mul__(self, rhs):
        """naive_poly_mul, ring_torus.rs:266-298"""
        if rhs.n != self.n:
            raise binding.FheError(binding.FHE_E_PARAM_MISMATCH, "Tn operands differ in n")
        return Tn(binding.tn_mul(self.n, self.coeffs, rhs.coeffs).reshape(self.coeffs.shape))

    def __eq__(self, other):
        return isinstance(other, Tn) and np.array_equal(self.coeffs, other.coeffs)


class TGLWE:
    """(a_0..a_{k-1}, b): `a` is (k, n), `b` is (n,) — or batches with a leading axis"""

    def __init__(self, a, b):
        self.a = np.ascontiguousarray(a, dtype=np.uint64)
        self.b = np.ascontiguousarray(b, dtype=np.uint64)

    def packed(self):
        return np.concatenate([self.a, self.b[..., None, :]], axis=-2)   # [.., k+1, n]

    def __mul__(self, plaintext):
        """plaintext multiplication, tglwe.rs:182-194: every component times the Tn"""
        x = self.packed()
        k = x.shape[-2] - 1
        out = binding.tglwe_mul_tn(plaintext.n, k, x, plaintext.coeffs).reshape(x.shape)
        return TGLWE(out[..., :k, :], out[..., k, :])


class TGLev:
    """l TGLWEs (tggsw.rs:65): rows [l][(k+1)][n]"""

    def __init__(self, rows):
        self.rows = np.ascontiguousarray(rows, dtype=np.uint64)

    def __mul__(self, v):
        """dot product with a Vec<Tn> (usually a decomposition), tggsw.rs:139-149"""
        l, k1, n = self.rows.shape
        if len(v) != l:
            raise binding.FheError(binding.FHE_E_INVALID, "TGLev * Vec<Tn>: lengths differ")   # assert_eq!, :143
        vv = np.stack([t.coeffs for t in v])
        out = binding.tglev_mul(n, k1 - 1, l, self.rows, vv)[0]
        return TGLWE(out[: k1 - 1], out[k1 - 1])


class TGGSW:
    """([k x TGLev], TGLev), each TGLev = l TGLWEs (tggsw.rs:12-14): rows[(k+1)][l][(k+1)][n]"""
    BETA, L = 2, 64   # hard-coded in the reference's external product (tggsw.rs:49-50)

    def __init__(self, rows):
        self.rows = np.ascontiguousarray(rows, dtype=np.uint64)
        k1, l, k1b, _ = self.rows.shape
        if k1 != k1b:
            raise ValueError("rows must be [(k+1)][l][(k+1)][n]")

    def __mul__(self, tglwe):
        """external product, tggsw.rs:45-62"""
        k1, l, _, n = self.rows.shape
        x = tglwe.packed()
        out = binding.tggsw_external_product(n, k1 - 1, l, self.rows, x).reshape(x.shape)
        return TGLWE(out[..., : k1 - 1, :], out[..., k1 - 1, :])
