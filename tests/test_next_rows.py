"""Rows N1/N2 of SURVEY.md §8f: BFV tensor + relinearise and the TFHE products, which the
reference computes by schoolbook loops (ring_n.rs:307-320, ring_torus.rs:266-298) and this
repo computes through multi-prime NTTs + CRT (fhe-study_amd/csrc/zring.hip).

CPU part: oracle/fhe_next_oracle.c against the reference's literal KATs and against the
reference's own PROPERTY tests (bfv/src/lib.rs:504-601), restated with a small BFV written in
Python integers here — that is what pins the reading of the i64-wrap / f64 semantics.
GPU part: word-for-word parity of the HIP path with that oracle."""
import math

import numpy as np
import pytest

from conftest import Q16, Q61

I64 = 1 << 63
U64 = 1 << 64


# ---- a tiny BFV over Python ints, following bfv/src/lib.rs line by line -------------------------

def _wrap_i64(x):
    x &= U64 - 1
    return x - U64 if x >= I64 else x


def _conv(a, b):
    """ring_n::naive_mul (ring_n.rs:307-320): linear convolution, `as i64`"""
    n = len(a)
    r = [0] * (2 * n - 1)
    for i in range(n):
        for j in range(n):
            r[i + j] += a[i] * b[j]
    return [_wrap_i64(x) for x in r]


def _rust_round(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def _from_f64(q, e):
    """Zq::from_f64 (zq.rs:32-39)"""
    e = _rust_round(e)
    e = max(min(e, I64 - 1), -I64)
    return e % q if (e < 0 or e >= q) else e


def _fold(q, z, n):
    """Rq::from_vec → modulus (ring_nq.rs:132-141)"""
    out = list(z[:n])
    for i in range(n, len(z)):
        out[i - n] = (out[i - n] - z[i]) % q
    return out


def _from_vec_i64(q, n, c):
    """Rq::from_vec_i64 (ring_nq.rs:164-170): `*c as f64` then from_f64, then fold"""
    return _fold(q, [_from_f64(q, float(x)) for x in c], n)


def _rq_mul(q, a, b):
    """exact product in Z_q[X]/(X^n+1) (what the NTT path returns)"""
    n = len(a)
    r = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                r[k] = (r[k] + a[i] * b[j]) % q
            else:
                r[k - n] = (r[k - n] - a[i] * b[j]) % q
    return r


def _mul_div_round_rq(q, a, num, den):
    """Ring::mul_div_round for Rq (ring_nq.rs:100-113)"""
    return [_from_f64(q, _rust_round((float(num) * float(v)) / float(den))) for v in a]


class MiniBFV:
    """new_key :118-139, encrypt :142-163, decrypt :165-181, rlk_key :202-225 of bfv/src/lib.rs"""

    def __init__(self, q, n, t, p, rng):
        self.q, self.n, self.t, self.p, self.rng = q, n, t, p, rng

    def _err(self, mod):
        return [_from_f64(mod, float(self.rng.normal(0, 3.2))) for _ in range(self.n)]

    def new_key(self):
        q, n = self.q, self.n
        s = [int(x) for x in self.rng.integers(0, 2, n)]
        a = [int(x) for x in self.rng.integers(0, q, n)]
        e = self._err(q)
        neg_a = [(-x) % q for x in a]
        pk0 = [(x + y) % q for x, y in zip(_rq_mul(q, neg_a, s), e)]
        return s, (pk0, a)

    def encrypt(self, pk, m):
        q, n, t = self.q, self.n, self.t
        u = [_from_f64(q, float(self.rng.uniform(-1, 1))) for _ in range(n)]
        e1, e2 = self._err(q), self._err(q)
        delta = q // t
        c0 = [(x + y + mm * delta) % q for x, y, mm in zip(_rq_mul(q, pk[0], u), e1, m)]
        c1 = [(x + y) % q for x, y in zip(_rq_mul(q, pk[1], u), e2)]
        return c0, c1

    def decrypt(self, s, c):
        q = self.q
        cs = [(x + y) % q for x, y in zip(c[0], _rq_mul(q, c[1], s))]
        return [v % self.t for v in _mul_div_round_rq(q, cs, self.t, q)]

    def rlk_key(self, s):
        pq, n, p = self.p * self.q, self.n, self.p
        a = [int(x) for x in self.rng.integers(0, pq, n)]
        e = self._err(pq)
        tmp = lambda x, y: _from_vec_i64(pq, n, _conv(x, y))  # tmp_naive_mul, lib.rs:92-97
        as_e = [(x + y) % pq for x, y in zip(tmp(a, s), e)]
        ss_p = [(x * (p % pq)) % pq for x in tmp(s, s)]
        return [((-x) % pq + y) % pq for x, y in zip(as_e, ss_p)], a


def test_ring_n_kat(oracle):
    """arith/src/ring_n.rs:453-470 (SageMath vectors): naive_mul + fold, n = 2"""
    q = Q16
    for a, want in (([q - 1, q - 1], [0, 8589934592]), ([1, q - 1], [-4294967295, 131072])):
        c = oracle.r_naive_mul(2, a, a)[0]
        assert c.tolist() == _conv(a, a)
        assert oracle.r_modulus(2, c).tolist() == want


def test_oracle_conv_wraps_like_i64(oracle):
    rng = np.random.default_rng(3)
    a = rng.integers(0, I64, 8, dtype=np.int64)
    b = rng.integers(0, I64, 8, dtype=np.int64)
    assert oracle.r_naive_mul(8, a, b)[0].tolist() == _conv([int(x) for x in a], [int(x) for x in b])


def test_oracle_mul_div_round_matches_python(oracle):
    rng = np.random.default_rng(4)
    n, q = 8, Q16
    v = np.concatenate([rng.integers(-I64, I64, 10, dtype=np.int64),
                        np.array([0, 1, -1, 32768, -32769, I64 - 1, -I64], dtype=np.int64)])[: 2 * n - 1]
    for num, den in ((2, q), (1, q * q), (7, 3)):
        z = [_from_f64(q, _rust_round((float(num) * float(int(x))) / float(den))) for x in v]
        assert oracle.mul_div_round(q, n, v, num, den)[0].tolist() == _fold(q, z, n)


def test_t64_decompose_recomposes(oracle):
    """arith/src/torus.rs:160-190 test_decompose: most significant bit first"""
    import ctypes
    x = np.array([12345, 0, U64 - 2, 0x8000000000000001], dtype=np.uint64)
    out = np.empty(64 * 4, dtype=np.uint64)
    p = ctypes.POINTER(ctypes.c_uint64)
    oracle.lib.oracle_tn_decompose(4, 64, x.ctypes.data_as(p), out.ctypes.data_as(p))
    d = out.reshape(64, 4)
    for j in range(4):
        acc = 0
        for i in range(64):
            acc = (acc << 1) | int(d[i, j])
        assert acc == int(x[j])


@pytest.mark.parametrize("seed", range(6))
def test_bfv_oracle_passes_the_reference_property_tests(oracle, seed):
    """bfv/src/lib.rs:504-601 test_tensor / test_mul_relin: n = 16, q = 2^16+1, t = 2, p = q^2.
    decrypt(RLWE::mul(c1, c2)) == m1*m2 mod (t, X^n+1), with the oracle doing tensor+relin."""
    q, n, t = Q16, 16, 2
    p = q * q
    rng = np.random.default_rng(100 + seed)
    bfv = MiniBFV(q, n, t, p, rng)
    for _ in range(8):
        s, pk = bfv.new_key()
        rlk0, rlk1 = bfv.rlk_key(s)
        m1 = [int(x) for x in rng.integers(0, t, n)]
        m2 = [int(x) for x in rng.integers(0, t, n)]
        c1, c2 = bfv.encrypt(pk, m1), bfv.encrypt(pk, m2)
        want = _rq_mul(t, m1, m2)
        # test_tensor: decrypt the non-relinearised triple with s and s^2
        ca, cb, cc = (x[0].tolist() for x in oracle.bfv_tensor(q, n, t, c1[0], c1[1], c2[0], c2[1]))
        s2 = _rq_mul(q, s, s)
        m3 = [(x + y + z) % q for x, y, z in zip(ca, _rq_mul(q, cb, s), _rq_mul(q, cc, s2))]
        assert [v % t for v in _mul_div_round_rq(q, m3, t, q)] == want
        # test_mul_relin
        o0, o1 = oracle.bfv_mul(q, n, t, p * q, rlk0, rlk1, c1[0], c1[1], c2[0], c2[1])
        assert bfv.decrypt(s, (o0[0].tolist(), o1[0].tolist())) == want


def test_external_product_oracle_gadget_identity(oracle):
    """With noiseless rows under the zero key (a-rows 0, b-row level d = m * (MAX >> (d+1)), as
    TGLev::encrypt_s scales, tggsw.rs:100-118), the external product body is
    m (*) sum_d g_d * digit_d(b) mod (2^64, X^n+1): an independent Python-int computation."""
    n, k, l = 8, 2, 64
    rng = np.random.default_rng(9)
    m = [int(x) for x in rng.integers(0, 16, n)]
    g = [(U64 - 1) >> (d + 1) if d + 1 < 64 else 1 for d in range(l)]
    tggsw = np.zeros((k + 1, l, k + 1, n), dtype=np.uint64)
    for d in range(l):
        tggsw[k, d, k] = np.array([(x * g[d]) % U64 for x in m], dtype=np.uint64)
    tglwe = rng.integers(0, U64, (1, k + 1, n), dtype=np.uint64)
    out = oracle.external_product(n, k, l, tggsw, tglwe)[0]
    b = [int(x) for x in tglwe[0, k]]
    y = [sum(g[d] * ((x >> (l - 1 - d)) & 1) for d in range(l)) % U64 for x in b]
    want = [0] * n
    for i in range(n):
        for j in range(n):
            kk = i + j
            if kk < n:
                want[kk] = (want[kk] + m[i] * y[j]) % U64
            else:
                want[kk - n] = (want[kk - n] - m[i] * y[j]) % U64
    assert out[k].tolist() == want
    assert not out[:k].any()


# ---- GPU parity ------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_r_naive_mul(pkg, oracle):
    B = pkg.binding
    q = Q16
    for a in ([q - 1, q - 1], [1, q - 1]):                       # ring_n.rs:453-470
        assert B.r_naive_mul(2, a, a)[0].tolist() == _conv(a, a)
    rng = np.random.default_rng(21)
    for n, hi in ((16, Q16), (256, Q16), (2048, Q61), (64, I64), (1024, I64)):   # 1, 2-3 and 3 primes; wraps
        a = rng.integers(0, hi, (3, n), dtype=np.int64)
        b = rng.integers(0, hi, (3, n), dtype=np.int64)
        assert np.array_equal(B.r_naive_mul(n, a, b), oracle.r_naive_mul(n, a, b)), (n, hi)
    a = np.full((1, 512), I64 - 1, dtype=np.int64)                # extreme magnitude
    assert np.array_equal(B.r_naive_mul(512, a, a), oracle.r_naive_mul(512, a, a))


@pytest.mark.gpu
def test_gpu_mul_div_round(pkg, oracle):
    import torch

    n, q = 64, Q16
    rng = np.random.default_rng(22)
    v = rng.integers(-I64, I64, (5, 2 * n), dtype=np.int64)
    v[0, :8] = [0, 1, -1, 32768, -32769, I64 - 1, -I64, 65537]
    v[:, 2 * n - 1] = 0
    dv = torch.from_numpy(v).cuda()
    for num, den in ((2, q), (1, q * q), (7, 3), (1, 1)):
        out = torch.empty((5, n), dtype=torch.int64, device="cuda")
        pkg.binding._check(pkg.load_library().fhe_mul_div_round_dev(q, n, dv.data_ptr(), num, den, out.data_ptr(), 5, None))
        torch.cuda.synchronize()
        want = oracle.mul_div_round(q, n, v[:, : 2 * n - 1], num, den)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want), (num, den)


@pytest.mark.gpu
@pytest.mark.parametrize("q,n,p,t,batch", [
    (Q16, 16, Q16 * Q16, 2, 64),          # the reference's own test parameters (lib.rs:506-511,559-564)
    (Q16, 512, Q16 * Q16, 2, 4),
    (Q16, 8192, Q16 * Q16, 2, 2),         # BASELINE.json configs[2]: N = 8192
    (Q61, 64, 2, 16, 3),                  # 61-bit q: three CRT primes, i64 wrap everywhere
    (12289, 1024, 12289, 3, 2),
    (Q16, 2048, Q16 * Q16, 2, 3),         # 2n = 4096: 256-thread workgroups of the small-prime kernels (bfv32.hip)
    (786433, 4096, 786433 ** 2, 5, 2),    # 20-bit q: the tensor fits two 27-bit primes, the relinearisation (20 + 59 + 12 bits) does
                                          # not fit three — small-prime tensor feeding the 61-bit relinearisation
])
def test_gpu_bfv_tensor_and_mul(pkg, oracle, q, n, p, t, batch):
    rng = np.random.default_rng(q % 1000 + n)
    a0, a1, b0, b1 = (rng.integers(0, q, (batch, n), dtype=np.uint64) for _ in range(4))
    rlk0, rlk1 = (rng.integers(0, p * q, n, dtype=np.uint64) for _ in range(2))
    got = pkg.binding.bfv_tensor(q, n, t, a0, a1, b0, b1)
    want = oracle.bfv_tensor(q, n, t, a0, a1, b0, b1)
    for g, w, name in zip(got, want, "c0 c1 c2".split()):
        assert np.array_equal(g, w), name
    g0, g1 = pkg.binding.bfv_mul(q, n, t, p * q, rlk0, rlk1, a0, a1, b0, b1)
    w0, w1 = oracle.bfv_mul(q, n, t, p * q, rlk0, rlk1, a0, a1, b0, b1)
    assert np.array_equal(g0, w0) and np.array_equal(g1, w1)


@pytest.mark.gpu
def test_gpu_bfv_mul_decrypts(pkg):
    """test_mul_relin (bfv/src/lib.rs:557-601) end to end with the HIP path doing RLWE::mul"""
    q, n, t = Q16, 16, 2
    p = q * q
    rng = np.random.default_rng(77)
    bfv = MiniBFV(q, n, t, p, rng)
    for _ in range(10):
        s, pk = bfv.new_key()
        rlk0, rlk1 = bfv.rlk_key(s)
        m1 = [int(x) for x in rng.integers(0, t, n)]
        m2 = [int(x) for x in rng.integers(0, t, n)]
        c1, c2 = bfv.encrypt(pk, m1), bfv.encrypt(pk, m2)
        o0, o1 = pkg.binding.bfv_mul(q, n, t, p * q, rlk0, rlk1, c1[0], c1[1], c2[0], c2[1])
        assert bfv.decrypt(s, (o0[0].tolist(), o1[0].tolist())) == _rq_mul(t, m1, m2)


@pytest.mark.gpu
def test_gpu_tn_mul(pkg, oracle):
    rng = np.random.default_rng(23)
    for n in (2, 4, 64, 1024, 4096):
        a = rng.integers(0, U64, (3, n), dtype=np.uint64)
        b = rng.integers(0, U64, (3, n), dtype=np.uint64)
        a[0] = U64 - 1
        b[0] = U64 - 1                                   # largest magnitude, worst case for the lift
        assert np.array_equal(pkg.binding.tn_mul(n, a, b), oracle.tn_mul(n, a, b)), n


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,l,batch", [(64, 4, 64, 3), (1024, 1, 8, 2), (8, 2, 3, 5)])
def test_gpu_tglwe_times_tn_and_tglev_times_vec(pkg, oracle, n, k, l, batch):
    """TGLWE x Tn (tglwe.rs:182-194) and TGLev x Vec<Tn> (tggsw.rs:139-149) against their definitions:
    single Tn products (the oracle's schoolbook mod 2^64) and wrapping sums"""
    rng = np.random.default_rng(n * 7 + l)
    c = rng.integers(0, U64, (batch, k + 1, n), dtype=np.uint64)
    p = rng.integers(0, U64, (batch, n), dtype=np.uint64)
    c[0] = U64 - 1
    p[0] = U64 - 1
    got = pkg.binding.tglwe_mul_tn(n, k, c, p)
    want = np.stack([oracle.tn_mul(n, c[b], np.broadcast_to(p[b], (k + 1, n))) for b in range(batch)])
    assert np.array_equal(got, want)

    tglev = rng.integers(0, U64, (l, k + 1, n), dtype=np.uint64)
    v = rng.integers(0, U64, (batch, l, n), dtype=np.uint64)
    v[0] = U64 - 1
    got = pkg.binding.tglev_mul(n, k, l, tglev, v)
    want = np.zeros((batch, k + 1, n), dtype=np.uint64)
    for b in range(batch):
        for d in range(l):
            want[b] += oracle.tn_mul(n, tglev[d], np.broadcast_to(v[b, d], (k + 1, n)))      # u64 add wraps
    assert np.array_equal(got, want)
    # the host mirror reads like the reference: TGLev * Vec<Tn>, TGLWE * Tn
    T = pkg.tfhe
    r = T.TGLev(tglev) * [T.Tn(v[1, d]) for d in range(l)]
    assert np.array_equal(r.packed(), want[1])
    r = T.TGLWE(c[1, :k], c[1, k]) * T.Tn(p[1])
    assert np.array_equal(r.packed(), pkg.binding.tglwe_mul_tn(n, k, c[1:2], p[1:2])[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,l,batch", [(64, 4, 64, 3),      # tfhe/src/tggsw.rs:157-196 test shape
                                         (1024, 1, 64, 2),     # BASELINE.json configs[3] shape
                                         (16, 2, 8, 5),
                                         (8, 1, 64, 2),        # n < 16: no single-pass digit kernel
                                         (16384, 1, 2, 1)])    # two-pass size: digits materialised once
def test_gpu_external_product(pkg, oracle, n, k, l, batch):
    rng = np.random.default_rng(n + k)
    tggsw = rng.integers(0, U64, (k + 1, l, k + 1, n), dtype=np.uint64)
    tglwe = rng.integers(0, U64, (batch, k + 1, n), dtype=np.uint64)
    got = pkg.binding.tggsw_external_product(n, k, l, tggsw, tglwe).reshape(batch, k + 1, n)
    assert np.array_equal(got, oracle.external_product(n, k, l, tggsw, tglwe))


@pytest.mark.gpu
def test_gpu_host_mirrors_read_like_the_reference(pkg, oracle):
    """RLWE::mul(t, &rlk, &a, &b) and `tgsw * tlwe` through the mirror classes"""
    from fhe_study_amd import bfv, tfhe

    q, n, t = Q16, 16, 2
    pq = q * q * q
    rng = np.random.default_rng(5)
    param = pkg.RingParam(q, n)
    ct = lambda: bfv.RLWE(pkg.Rq(param, rng.integers(0, q, n, dtype=np.uint64)),
                          pkg.Rq(param, rng.integers(0, q, n, dtype=np.uint64)))
    a, b = ct(), ct()
    rlk = bfv.RLK(rng.integers(0, pq, n, dtype=np.uint64), rng.integers(0, pq, n, dtype=np.uint64), pq)
    c3 = bfv.RLWE.mul(t, rlk, a, b)
    w0, w1 = oracle.bfv_mul(q, n, t, pq, rlk.rlk0, rlk.rlk1, a.c0.coeffs, a.c1.coeffs, b.c0.coeffs, b.c1.coeffs)
    assert np.array_equal(c3.c0.coeffs, w0[0]) and np.array_equal(c3.c1.coeffs, w1[0])

    n, k, l = 64, 4, 64
    tgsw = tfhe.TGGSW(rng.integers(0, U64, (k + 1, l, k + 1, n), dtype=np.uint64))
    tlwe = tfhe.TGLWE(rng.integers(0, U64, (k, n), dtype=np.uint64), rng.integers(0, U64, n, dtype=np.uint64))
    res = tgsw * tlwe
    want = oracle.external_product(n, k, l, tgsw.rows, tlwe.packed())[0]
    assert np.array_equal(res.a, want[:k]) and np.array_equal(res.b, want[k])
    x, y = tfhe.Tn(rng.integers(0, U64, n, dtype=np.uint64)), tfhe.Tn(rng.integers(0, U64, n, dtype=np.uint64))
    assert np.array_equal((x * y).coeffs, oracle.tn_mul(n, x.coeffs, y.coeffs)[0])


@pytest.mark.gpu
def test_gpu_reducing_loads_at_every_transform_size(pkg, oracle):
    """the operand reduction/padding fused into the forward load (SRC_REDUCE / RSRC kernels) exists
    per transform size: Tn x Tn (no padding) runs it at n, naive_mul (zero-padding) at 2n — cover every
    single-pass size 16..8192 and the three strided shapes (2^14, 2^15, >= 2^16)"""
    B = pkg.binding
    rng = np.random.default_rng(31)
    for n in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        batch = 2 if n <= 8192 else 1
        a = rng.integers(0, U64, (batch, n), dtype=np.uint64)
        b = rng.integers(0, U64, (batch, n), dtype=np.uint64)
        assert np.array_equal(B.tn_mul(n, a, b), oracle.tn_mul(n, a, b)), n
    for n in (8, 16, 128, 4096, 8192, 16384, 32768):            # transforms of 2n = 16 .. 65536
        a = rng.integers(0, Q61, (1, n), dtype=np.int64)
        b = rng.integers(0, Q61, (1, n), dtype=np.int64)
        assert np.array_equal(B.r_naive_mul(n, a, b), oracle.r_naive_mul(n, a, b)), n


@pytest.mark.gpu
@pytest.mark.parametrize("n", [32, 128, 256, 512, 2048, 4096, 8192])
def test_gpu_external_product_digit_load_at_every_single_pass_size(pkg, oracle, n):
    """the bit-extracting forward load (SRC_DIGITS) is instantiated per size: the sizes the shapes
    of test_gpu_external_product do not touch, with l < 64 and a ragged polynomial count"""
    k, l, batch = 1, 3, 3
    rng = np.random.default_rng(n)
    tggsw = rng.integers(0, U64, (k + 1, l, k + 1, n), dtype=np.uint64)
    tglwe = rng.integers(0, U64, (batch, k + 1, n), dtype=np.uint64)
    got = pkg.binding.tggsw_external_product(n, k, l, tggsw, tglwe).reshape(batch, k + 1, n)
    assert np.array_equal(got, oracle.external_product(n, k, l, tggsw, tglwe))
