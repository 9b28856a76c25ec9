"""Rows N3/N4 of SURVEY.md §8f: the reference's batch surfaces (TR.TR, TR x R / GLWE x R,
GLev x Vec<R>, GLWE::key_switch) and element-wise Rq glue, device-resident.  CPU part pins the
oracle (oracle/fhe_glue_oracle.c) against Python-int restatements and the reference's unit tests;
GPU part is word-for-word parity of the HIP path with the oracle."""
import math

import numpy as np
import pytest

from conftest import Q16, Q61

U64 = 1 << 64
Q63 = 9223372036844421121      # the largest prime = 1 (mod 2^17) below 2^63: the top of the reference's range (zq.rs:225)


def _rq_mul(q, a, b):
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


def _u(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


# ---- CPU: the oracle itself ---------------------------------------------------------------------

def test_oracle_zq_decompose_matches_reference_tests(oracle):
    """arith/src/ring_nq.rs:706-730 test_rq_decompose: q = 16, n = 4, beta = 4, l = 2"""
    a = _u([7, 14, 3, 6])
    out = np.empty((2, 4), dtype=np.uint64)
    oracle.glue("rq_decompose", 16, 4, a, 4, 2, out)
    assert out[0].tolist() == [1, 3, 0, 1] and out[1].tolist() == [3, 2, 3, 2]


def test_oracle_base2_decompose_recomposes(oracle):
    """arith/src/zq.rs:374-435 (test_decompose): beta = 2, values below 2^l recompose exactly"""
    q, l, n = Q16, 16, 8
    a = _u([0, 1, 2, 3, 12345, 65535, 40000, 7])
    out = np.empty((l, n), dtype=np.uint64)
    oracle.glue("rq_decompose", q, n, a, 2, l, out)
    rec = sum(int(1 << (l - 1 - d)) * out[d].astype(object) for d in range(l))
    assert [int(x) for x in rec] == a.tolist()
    # saturation branch (zq.rs:176-180): v >= 2^l -> all ones
    oracle.glue("rq_decompose", q, n, _u([65536] * n), 2, l, out)
    assert (out == 1).all()


def test_oracle_elementwise_against_python(oracle):
    q, n = Q61, 16
    rng = np.random.default_rng(1)
    a, b = (rng.integers(0, q, n, dtype=np.uint64) for _ in range(2))
    a[0], b[0] = 0, 0
    out = np.empty(n, dtype=np.uint64)
    oracle.glue("rq_add", q, n, a, b, out); assert out.tolist() == [(int(x) + int(y)) % q for x, y in zip(a, b)]
    oracle.glue("rq_sub", q, n, a, b, out); assert out.tolist() == [(int(x) - int(y)) % q for x, y in zip(a, b)]
    oracle.glue("rq_neg", q, n, a, out); assert out.tolist() == [(-int(x)) % q for x in a]
    oracle.glue("rq_mul_by_u64", q, n, a, (1 << 64) - 3, out)
    assert out.tolist() == [(int(x) * (((1 << 64) - 3) % q)) % q for x in a]
    p = 1 << 20
    oracle.glue("rq_mod_switch", q, n, a, p, out)
    rr = lambda x: math.floor(x + 0.5)
    assert out.tolist() == [int(rr((float(int(x)) * float(p)) / float(q))) % p for x in a]
    # remodule / mul_by_f64 / div_round (ring_nq.rs:82-88,282-292,299-306) against Python floats
    oracle.glue("rq_remodule", n, a, 65537, out); assert out.tolist() == [int(x) % 65537 for x in a]
    small = (a % np.uint64(1 << 40)).astype(np.uint64)         # keeps v*s inside i64: no saturation
    rnd = lambda x: math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)
    oracle.glue("rq_mul_by_f64", q, n, small, -2.5, out)
    assert out.tolist() == [int(rnd(float(int(x)) * -2.5)) % q for x in small]
    oracle.glue("rq_div_round", q, n, a, 1000, out)
    assert out.tolist() == [int(rnd(float(int(x)) / 1000.0)) % q for x in a]


def test_oracle_batch_surfaces_against_python(oracle):
    q, n, k, l = Q16, 8, 3, 4
    rng = np.random.default_rng(2)
    a = rng.integers(0, q, (k, n), dtype=np.uint64)
    b = rng.integers(0, q, (k, n), dtype=np.uint64)
    c = np.empty(n, dtype=np.uint64)
    oracle.glue("tr_dot", q, n, k, a, b, c)
    want = [0] * n
    for i in range(k):
        want = [(x + y) % q for x, y in zip(want, _rq_mul(q, [int(v) for v in a[i]], [int(v) for v in b[i]]))]
    assert c.tolist() == want
    glev = rng.integers(0, q, (l, k + 1, n), dtype=np.uint64)
    v = rng.integers(0, q, (l, n), dtype=np.uint64)
    out = np.empty((k + 1, n), dtype=np.uint64)
    oracle.glue("glev_mul", q, n, k, l, glev, v, out)
    for cc in range(k + 1):
        w = [0] * n
        for d in range(l):
            w = [(x + y) % q for x, y in zip(w, _rq_mul(q, [int(t) for t in glev[d, cc]], [int(t) for t in v[d]]))]
        assert out[cc].tolist() == w


# ---- GPU parity ----------------------------------------------------------------------------------

def _dev(x):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("q,n", [(Q16, 64), (Q61, 1024), (Q63, 256)])
def test_gpu_elementwise(pkg, oracle, q, n):
    import torch

    L, chk = pkg.load_library(), pkg.binding._check
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(n)
    batch = 5
    a = rng.integers(0, q, (batch, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, n), dtype=np.uint64)
    a[0, :4] = [0, q - 1, 1, q // 2]
    b[0, :4] = [0, q - 1, q - 1, q // 2 + 1]
    da, db = _dev(a), _dev(b)
    dc = torch.empty_like(da)
    want = np.empty_like(a)

    def cmp(name, *args):
        torch.cuda.synchronize()
        for i in range(batch):
            oracle.glue(name, *[(x[i] if isinstance(x, np.ndarray) and x.ndim == 2 else x) for x in args][:-1], want[i])
        assert np.array_equal(_host(dc), want), name

    chk(L.fhe_rq_add_dev(plan.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, None)); cmp("rq_add", q, n, a, b, None)
    chk(L.fhe_rq_sub_dev(plan.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, None)); cmp("rq_sub", q, n, a, b, None)
    chk(L.fhe_rq_neg_dev(plan.handle, da.data_ptr(), dc.data_ptr(), batch, None)); cmp("rq_neg", q, n, a, None)
    for s in (0, 1, q - 1, q + 5, (1 << 64) - 1):
        chk(L.fhe_rq_mul_by_u64_dev(plan.handle, da.data_ptr(), s, dc.data_ptr(), batch, None)); cmp("rq_mul_by_u64", q, n, a, s, None)
    for p in (2, 1 << 10, q - 2, min(q * 3, (1 << 63) - 1)):
        chk(L.fhe_rq_mod_switch_dev(q, p, da.data_ptr(), dc.data_ptr(), batch * n, None)); cmp("rq_mod_switch", q, n, a, p, None)
    for num, den in ((2, q), (16, q), (1, 3)):
        chk(L.fhe_rq_mul_div_round_dev(q, num, den, da.data_ptr(), dc.data_ptr(), batch * n, None)); cmp("rq_mul_div_round", q, n, a, num, den, None)
    for p in (2, 65537, q, (1 << 63) - 25):
        chk(L.fhe_rq_remodule_dev(p, da.data_ptr(), dc.data_ptr(), batch * n, None)); cmp("rq_remodule", n, a, p, None)
    for s in (0.0, 1.0, -2.5, 0.333, 1e6, -1e30):                 # the last saturates `as i64`
        chk(L.fhe_rq_mul_by_f64_dev(q, s, da.data_ptr(), dc.data_ptr(), batch * n, None)); cmp("rq_mul_by_f64", q, n, a, s, None)
    for s in (1, 2, 1000, q - 1, (1 << 64) - 1):
        chk(L.fhe_rq_div_round_dev(q, s, da.data_ptr(), dc.data_ptr(), batch * n, None)); cmp("rq_div_round", q, n, a, s, None)


@pytest.mark.gpu
@pytest.mark.parametrize("q,n,beta,l", [(16, 4, 4, 2), (Q16, 64, 2, 16), (Q16, 64, 2, 8), (Q16, 32, 4, 8), (Q61, 256, 2, 61), (Q63, 64, 2, 63), (Q63, 64, 4, 12)])
def test_gpu_decompose(pkg, oracle, q, n, beta, l):
    import torch

    rng = np.random.default_rng(l)
    rows = 3
    a = rng.integers(0, q, (rows, n), dtype=np.uint64)
    if q == 16:
        a[0] = [7, 14, 3, 6]                      # ring_nq.rs:706-730
    da = _dev(a)
    dout = torch.empty((rows, l, n), dtype=torch.int64, device="cuda")
    pkg.binding._check(pkg.load_library().fhe_rq_decompose_dev(q, n, beta, l, da.data_ptr(), dout.data_ptr(), rows, None))
    torch.cuda.synchronize()
    want = np.empty((rows, l, n), dtype=np.uint64)
    for r in range(rows):
        oracle.glue("rq_decompose", q, n, a[r], beta, l, want[r])
    assert np.array_equal(_host(dout), want)


@pytest.mark.gpu
@pytest.mark.parametrize("q,n,k,batch", [(Q16, 8, 16, 3), (Q16, 128, 16, 2), (Q61, 4096, 2, 3), (Q61, 16384, 3, 2), (Q63, 256, 5, 3), (Q63, 16384, 2, 2)])
def test_gpu_tr_dot_and_mul_r(pkg, oracle, q, n, k, batch):
    """TR.TR and GLWE x R at the reference's test shapes (gfhe/src/glwe.rs tests: k = 16, n = 8..128)"""
    import torch

    L, chk = pkg.load_library(), pkg.binding._check
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(k)
    a = rng.integers(0, q, (batch, k, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, k, n), dtype=np.uint64)
    p = rng.integers(0, q, (batch, n), dtype=np.uint64)
    da, db, dp = _dev(a), _dev(b), _dev(p)
    dc = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    chk(L.fhe_tr_dot_dev(plan.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), k, batch, 0, None))
    dout = torch.empty_like(da)
    chk(L.fhe_tr_mul_r_dev(plan.handle, da.data_ptr(), dp.data_ptr(), dout.data_ptr(), k, batch, 0, None))
    torch.cuda.synchronize()
    wc = np.empty((batch, n), dtype=np.uint64)
    wo = np.empty_like(a)
    for i in range(batch):
        oracle.glue("tr_dot", q, n, k, a[i], b[i], wc[i])
        oracle.glue("tr_mul_r", q, n, k, a[i], p[i], wo[i])
    assert np.array_equal(_host(dc), wc)
    assert np.array_equal(_host(dout), wo)

    # operands kept in the NTT domain (the generalised Rq.evals): same words, and a result left
    # there equals the forward transform of the coefficient result
    B = pkg.binding
    dA, dB, dP = (torch.empty_like(x) for x in (da, db, dp))
    plan.forward_dev(da.data_ptr(), dA.data_ptr(), batch * k)
    plan.forward_dev(db.data_ptr(), dB.data_ptr(), batch * k)
    plan.forward_dev(dp.data_ptr(), dP.data_ptr(), batch)
    dc2, dout2 = torch.empty_like(dc), torch.empty_like(dout)
    chk(L.fhe_tr_dot_dev(plan.handle, dA.data_ptr(), db.data_ptr(), dc2.data_ptr(), k, batch, B.FHE_A_IS_EVALS, None))
    assert np.array_equal(_host(dc2), wc)
    chk(L.fhe_tr_dot_dev(plan.handle, dA.data_ptr(), dB.data_ptr(), dc2.data_ptr(), k, batch,
                         B.FHE_A_IS_EVALS | B.FHE_B_IS_EVALS | B.FHE_OUT_EVALS, None))
    assert np.array_equal(_host(dc2), oracle.ntt(q, n, wc))
    chk(L.fhe_tr_mul_r_dev(plan.handle, da.data_ptr(), dP.data_ptr(), dout2.data_ptr(), k, batch, B.FHE_B_IS_EVALS, None))
    assert np.array_equal(_host(dout2), wo)
    chk(L.fhe_tr_mul_r_dev(plan.handle, dA.data_ptr(), dp.data_ptr(), dout2.data_ptr(), k, batch,
                           B.FHE_A_IS_EVALS | B.FHE_OUT_EVALS, None))
    assert np.array_equal(_host(dout2), oracle.ntt(q, n, wo.reshape(-1, n)).reshape(wo.shape))


@pytest.mark.gpu
@pytest.mark.parametrize("q,n,k,beta,l,batch", [(Q16, 128, 16, 2, 16, 2),     # gfhe/src/glwe.rs:582-594 test_key_switch
                                               (Q16, 16, 2, 4, 8, 3),
                                               (Q61, 1024, 2, 2, 61, 2),
                                               (Q63, 512, 2, 2, 63, 2),     # strict accumulators (two terms per fold)
                                               (Q63, 64, 1, 4, 9, 3)])
def test_gpu_glev_mul_and_key_switch(pkg, oracle, q, n, k, beta, l, batch):
    import torch

    L, chk = pkg.load_library(), pkg.binding._check
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(q % 97 + l)
    glev = rng.integers(0, q, (l, k + 1, n), dtype=np.uint64)
    v = rng.integers(0, q, (batch, l, n), dtype=np.uint64)
    dout = torch.empty((batch, k + 1, n), dtype=torch.int64, device="cuda")
    dglev, dv = _dev(glev), _dev(v)      # keep the device copies alive across the asynchronous call
    chk(L.fhe_glev_mul_dev(plan.handle, k, l, dglev.data_ptr(), dv.data_ptr(), dout.data_ptr(), batch, 0, None))
    torch.cuda.synchronize()
    want = np.empty((batch, k + 1, n), dtype=np.uint64)
    for i in range(batch):
        oracle.glue("glev_mul", q, n, k, l, glev, v[i], want[i])
    assert np.array_equal(_host(dout), want)
    # the key transformed once and reused (FHE_A_IS_EVALS), then both operands resident
    B = pkg.binding
    dGLEV, dV = torch.empty_like(dglev), torch.empty_like(dv)
    plan.forward_dev(dglev.data_ptr(), dGLEV.data_ptr(), l * (k + 1))
    plan.forward_dev(dv.data_ptr(), dV.data_ptr(), batch * l)
    dout.zero_()
    chk(L.fhe_glev_mul_dev(plan.handle, k, l, dGLEV.data_ptr(), dv.data_ptr(), dout.data_ptr(), batch, B.FHE_A_IS_EVALS, None))
    assert np.array_equal(_host(dout), want)
    chk(L.fhe_glev_mul_dev(plan.handle, k, l, dGLEV.data_ptr(), dV.data_ptr(), dout.data_ptr(), batch,
                           B.FHE_A_IS_EVALS | B.FHE_B_IS_EVALS | B.FHE_OUT_EVALS, None))
    assert np.array_equal(_host(dout), oracle.ntt(q, n, want.reshape(-1, n)).reshape(want.shape))

    glwe = rng.integers(0, q, (batch, k + 1, n), dtype=np.uint64)
    glwe[0, 0, :4] = [0, 1, (1 << min(l, 62)) % q, q - 1]           # both decompose branches
    ksk = rng.integers(0, q, (k, l, k + 1, n), dtype=np.uint64)
    dglwe, dksk = _dev(glwe), _dev(ksk)
    chk(L.fhe_glwe_key_switch_dev(plan.handle, k, beta, l, dglwe.data_ptr(), dksk.data_ptr(), dout.data_ptr(), batch, 0, None))
    dKSK, dout_k = torch.empty_like(dksk), torch.empty_like(dout)
    plan.forward_dev(dksk.data_ptr(), dKSK.data_ptr(), k * l * (k + 1))
    chk(L.fhe_glwe_key_switch_dev(plan.handle, k, beta, l, dglwe.data_ptr(), dKSK.data_ptr(), dout_k.data_ptr(), batch,
                                  B.FHE_A_IS_EVALS, None))
    torch.cuda.synchronize()
    assert torch.equal(dout, dout_k)                     # pre-transformed key switching key: same words
    for i in range(batch):
        oracle.glue("key_switch", q, n, k, beta, l, glwe[i], ksk, want[i])
    assert np.array_equal(_host(dout), want)


@pytest.mark.gpu
def test_gpu_host_buffer_forms_of_the_batch_surfaces(pkg, oracle):
    """fhe_tr_dot / fhe_tr_mul_r / fhe_glev_mul / fhe_glwe_key_switch take host buffers (what a
    shim of gfhe binds); same words as the oracle, batch > 1"""
    import ctypes
    L, chk = pkg.load_library(), pkg.binding._check
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    q, n, k, beta, l, batch = Q16, 64, 3, 2, 16, 2
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(5)
    a = rng.integers(0, q, (batch, k, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, k, n), dtype=np.uint64)
    p = rng.integers(0, q, (batch, n), dtype=np.uint64)
    c, wc = np.empty((batch, n), dtype=np.uint64), np.empty((batch, n), dtype=np.uint64)
    out, wo = np.empty_like(a), np.empty_like(a)
    chk(L.fhe_tr_dot(plan.handle, vp(a), vp(b), vp(c), k, batch))
    chk(L.fhe_tr_mul_r(plan.handle, vp(a), vp(p), vp(out), k, batch))
    for i in range(batch):
        oracle.glue("tr_dot", q, n, k, a[i], b[i], wc[i])
        oracle.glue("tr_mul_r", q, n, k, a[i], p[i], wo[i])
    assert np.array_equal(c, wc) and np.array_equal(out, wo)

    glev = rng.integers(0, q, (l, k + 1, n), dtype=np.uint64)
    v = rng.integers(0, q, (batch, l, n), dtype=np.uint64)
    glwe = rng.integers(0, q, (batch, k + 1, n), dtype=np.uint64)
    ksk = rng.integers(0, q, (k, l, k + 1, n), dtype=np.uint64)
    got, want = np.empty((batch, k + 1, n), dtype=np.uint64), np.empty((batch, k + 1, n), dtype=np.uint64)
    chk(L.fhe_glev_mul(plan.handle, k, l, vp(glev), vp(v), vp(got), batch))
    for i in range(batch):
        oracle.glue("glev_mul", q, n, k, l, glev, v[i], want[i])
    assert np.array_equal(got, want)
    chk(L.fhe_glwe_key_switch(plan.handle, k, beta, l, vp(glwe), vp(ksk), vp(got), batch))
    for i in range(batch):
        oracle.glue("key_switch", q, n, k, beta, l, glwe[i], ksk, want[i])
    assert np.array_equal(got, want)
