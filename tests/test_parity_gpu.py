"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle,
the committed golden fixtures and the reference's own test cases.  Bit-exact:
this is integer work, every comparison is word-for-word."""
import hashlib

import numpy as np
import pytest

from conftest import Q16, Q61, golden_cases, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(pkg):
    # fail loudly rather than skip: on the GPU box the native library must be the thing under test
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"


# ---- the reference's own tests, restated over the boundary ------------------------

def test_ref_test_ntt(pkg):
    """arith/src/ntt.rs:194-215 test_ntt"""
    param = pkg.RingParam(q=2**16 + 1, n=4)
    a = pkg.Rq.from_vec_u64(param, [1, 2, 3, 4])
    a_ntt = pkg.NTT.ntt(a)
    a_intt = pkg.NTT.intt(a_ntt)
    assert a == a_intt
    assert a_ntt.coeffs.tolist() == [7489, 56514, 17185, 49890]


def test_ref_test_ntt_loop(pkg):
    """arith/src/ntt.rs:217-234 test_ntt_loop: n=512, 1000 random polynomials (as one batch)"""
    param = pkg.RingParam(q=2**16 + 1, n=512)
    rng = np.random.default_rng(2024)
    a = pkg.Rq(param, rng.integers(0, param.q, size=(1000, 512), dtype=np.uint64))
    assert pkg.NTT.intt(pkg.NTT.ntt(a)) == a


def test_ref_test_mul(pkg, kat):
    """arith/src/ring_nq.rs:667-704 test_mul (through mul_mut, like the reference)"""
    for v in kat["ring_nq_test_mul"]:
        param = pkg.RingParam(v["q"], v["n"])
        a = pkg.Rq.from_vec_u64(param, v["a"])
        b = pkg.Rq.from_vec_u64(param, v["b"])
        expected_c = pkg.Rq.from_vec_u64(param, v["c"])
        c = pkg.mul_mut(a, b)
        assert c == expected_c
        # mul_mut stores the operands' evals (ring_nq.rs:568-573); the product carries its own (:582)
        assert a.evals is not None and b.evals is not None and c.evals is not None
        assert pkg.NTT.intt(pkg.Rq(param, c.evals)) == expected_c


def test_sage_pair(pkg, kat):
    v = kat["sage_ring_inputs"]
    param = pkg.RingParam(v["q"], v["n"])
    c = pkg.Rq.from_vec_u64(param, v["a"]) * pkg.Rq.from_vec_u64(param, v["b"])
    assert c.coeffs.tolist() == [65530, 6, 17, 24]


# ---- golden fixtures --------------------------------------------------------------

@pytest.mark.parametrize("name", golden_cases())
def test_golden(pkg, name):
    g = load_golden(name)
    plan = pkg.Plan(int(g["q"]), int(g["n"]))
    shape = g["a"].shape
    assert np.array_equal(plan.forward(g["a"]).reshape(shape), g["ntt_a"])
    assert np.array_equal(plan.forward(g["b"]).reshape(shape), g["ntt_b"])
    assert np.array_equal(plan.inverse(g["ntt_a"]).reshape(shape), g["a"])
    assert np.array_equal(plan.inverse(g["c_evals"]).reshape(shape), g["c"])
    c, ce, ae, be = plan.rq_mul(g["a"], g["b"])
    assert np.array_equal(c.reshape(shape), g["c"])
    assert np.array_equal(ce.reshape(shape), g["c_evals"])
    assert np.array_equal(ae.reshape(shape), g["ntt_a"])
    assert np.array_equal(be.reshape(shape), g["ntt_b"])
    assert np.array_equal(plan.pointwise_mul(g["ntt_a"], g["ntt_b"]).reshape(shape), g["c_evals"])


def test_golden_digests(pkg, oracle, kat):
    for d in kat["digests"].values():
        q, n = d["q"], d["n"]
        a = oracle.fill_synthetic(q, d["seed"], 0, n)
        A = pkg.Plan(q, n).forward(a)
        assert hashlib.sha256(A.tobytes()).hexdigest() == d["ntt_a_sha256"]
        assert A[:8].tolist() == d["ntt_a_head"]


# ---- every size, both moduli, against the oracle -----------------------------------

def _sizes(q):
    out = []
    n = 2
    while n <= (1 << 17) and (q - 1) % (2 * n) == 0:
        out.append(n)
        n *= 2
    return out


@pytest.mark.parametrize("q", [Q16, Q61])
def test_all_sizes_forward_inverse_mul(pkg, oracle, q):
    for n in _sizes(q):
        batch = 5 if n <= 8192 else 3          # ragged vs every workgroup packing (W = 256..1)
        a = oracle.fill_synthetic(q, 100 + n, 0, batch * n)
        b = oracle.fill_synthetic(q, 200 + n, 0, batch * n)
        plan = pkg.Plan(q, n)
        A = plan.forward(a)
        assert np.array_equal(A, oracle.ntt(q, n, a)), f"forward q={q} n={n}"
        assert np.array_equal(plan.inverse(A), a), f"round trip q={q} n={n}"
        assert np.array_equal(plan.inverse(b), oracle.intt(q, n, b)), f"inverse q={q} n={n}"
        c, ce, ae, be = plan.rq_mul(a, b)
        oc, oce, oae, obe = oracle.rq_mul(q, n, a, b)
        assert np.array_equal(c, oc) and np.array_equal(ce, oce), f"mul q={q} n={n}"
        assert np.array_equal(ae, oae) and np.array_equal(be, obe), f"mul evals q={q} n={n}"


@pytest.mark.parametrize("q,n", [(Q16, 16), (Q16, 256), (Q61, 64), (Q61, 512), (Q61, 2048)])
def test_product_equals_schoolbook(pkg, oracle, q, n):
    """the reference's independent cross-check, gfhe/src/glwe.rs:493-527"""
    a = oracle.fill_synthetic(q, 31, 0, 4 * n)
    b = oracle.fill_synthetic(q, 32, 0, 4 * n)
    c = pkg.Plan(q, n).rq_mul(a, b, want_evals=False)[0]
    assert np.array_equal(c, oracle.naive_negacyclic_mul(q, n, a, b))


@pytest.mark.parametrize("q", [Q16, Q61, 12289, 4611686018425815041])
def test_other_moduli_and_extreme_values(pkg, oracle, q):
    """q = 12289 (tiny), q = 2^62 - 2^19*3 + 1 just under the 2^62 engine limit; inputs of all
    zeros, all q-1 (maximum), unit vectors — the lazy-reduction bounds are tight there."""
    for n in (8, 256, 4096, 16384):
        if (q - 1) % (2 * n):
            continue
        rows = [np.zeros(n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64)]
        e = np.zeros(n, dtype=np.uint64); e[0] = 1; rows.append(e)
        e = np.zeros(n, dtype=np.uint64); e[n - 1] = q - 1; rows.append(e)
        rows.append(oracle.fill_synthetic(q, 77, 0, n))
        a = np.stack(rows)
        b = a[::-1].copy()
        plan = pkg.Plan(q, n)
        assert np.array_equal(plan.forward(a).reshape(-1), oracle.ntt(q, n, a).reshape(-1))
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1))
        assert np.array_equal(plan.rq_mul(a, b)[0].reshape(-1), oracle.rq_mul(q, n, a, b)[0].reshape(-1))
        assert (plan.forward(a) < q).all() and (plan.inverse(a) < q).all()


def test_modulus_just_below_engine_limit_exists():
    q = 4611686018425815041  # 0x3fffffffffe80001: prime, q-1 divisible by 2^18
    assert q < (1 << 62) and (q - 1) % (1 << 18) == 0


# ---- batch edge cases ---------------------------------------------------------------

def test_empty_and_ragged_batches(pkg, oracle):
    plan = pkg.Plan(Q61, 256)
    assert plan.forward(np.zeros(0, dtype=np.uint64)).size == 0
    for batch in (1, 15, 16, 17, 33):           # W = 16 units per workgroup at n=256
        a = oracle.fill_synthetic(Q61, batch, 0, batch * 256)
        assert np.array_equal(plan.forward(a), oracle.ntt(Q61, 256, a))
        assert np.array_equal(plan.inverse(a), oracle.intt(Q61, 256, a))
    plan = pkg.Plan(Q61, 16)
    for batch in (1, 255, 256, 257):            # W = 256 at n=16
        a = oracle.fill_synthetic(Q61, batch, 0, batch * 16)
        assert np.array_equal(plan.forward(a), oracle.ntt(Q61, 16, a))


def test_cached_evals_semantics(pkg, oracle):
    """ring_nq.rs:586-607: operands that carry evals are not transformed again"""
    q, n = Q61, 1024
    a = oracle.fill_synthetic(q, 1, 0, 2 * n)
    b = oracle.fill_synthetic(q, 2, 0, 2 * n)
    plan = pkg.Plan(q, n)
    A, B = plan.forward(a), plan.forward(b)
    want = oracle.rq_mul(q, n, a, b)[0]
    assert np.array_equal(plan.rq_mul(A, b, a_is_evals=True)[0], want)
    assert np.array_equal(plan.rq_mul(a, B, b_is_evals=True)[0], want)
    assert np.array_equal(plan.rq_mul(A, B, a_is_evals=True, b_is_evals=True)[0], want)
    # BFV-style reuse (bfv/src/lib.rs:132,169): s.compute_evals() once, then many products
    param = pkg.RingParam(q, n)
    s = pkg.Rq(param, a[:n]); s.compute_evals()
    for k in range(3):
        u = pkg.Rq(param, oracle.fill_synthetic(q, 50 + k, 0, n))
        assert np.array_equal((s * u).coeffs, oracle.rq_mul(q, n, a[:n], u.coeffs)[0])


def test_check_canonical(pkg):
    plan = pkg.Plan(Q16, 8)
    plan.check_canonical(np.arange(8, dtype=np.uint64))
    with pytest.raises(pkg.FheError) as ei:
        plan.check_canonical(np.array([0, 1, 2, Q16, 4, 5, 6, 7], dtype=np.uint64))
    assert ei.value.code == pkg.binding.FHE_E_NOT_CANONICAL


# ---- device-pointer entry points (what bench.py and downstream fusers use) -----------

def test_device_entry_points_in_place_and_tiled(pkg, oracle):
    import torch

    q, n, batch = Q61, 16384, 37
    dev = torch.device("cuda:0")
    plan = pkg.Plan(q, n)
    x = torch.empty(batch * n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    pkg.binding.fill_synthetic_dev(q, 0xABC, 0, batch * n, x.data_ptr(), st)
    a = x.cpu().numpy().view(np.uint64)
    assert np.array_equal(a, oracle.fill_synthetic(q, 0xABC, 0, batch * n))
    want = oracle.ntt(q, n, a)
    for tile in (0, 1, 5, 16, 64):              # two-pass sizes walk the batch in tiles
        pkg.binding.set_batch_tile(tile)
        y = x.clone()
        plan.forward_dev(y.data_ptr(), y.data_ptr(), batch, st)      # in place
        assert np.array_equal(y.cpu().numpy().view(np.uint64), want), f"tile={tile}"
        z = torch.empty_like(x)
        plan.inverse_dev(y.data_ptr(), z.data_ptr(), batch, st)      # out of place
        assert torch.equal(z, x), f"tile={tile}"
    pkg.binding.set_batch_tile(0)
    # Rq multiply on device with library workspace, keeping C evals
    b = torch.empty_like(x)
    pkg.binding.fill_synthetic_dev(q, 0xDEF, 0, batch * n, b.data_ptr(), st)
    c = torch.empty_like(x); ce = torch.empty_like(x)
    plan.rq_mul_dev(x.data_ptr(), b.data_ptr(), c.data_ptr(), batch, d_c_evals=ce.data_ptr(), stream=st)
    oc, oce, _, _ = oracle.rq_mul(q, n, a, b.cpu().numpy().view(np.uint64))
    assert np.array_equal(c.cpu().numpy().view(np.uint64), oc)
    assert np.array_equal(ce.cpu().numpy().view(np.uint64), oce)


# ---- BASELINE.json full-size configs through size-independent properties ---------------

def test_config2_n4096_batch4096_roundtrip_and_subset_parity(pkg, oracle):
    """configs[1]: N=4096, batch=4096 (128 MiB): forward+inverse bit-exact round trip,
    forward == oracle on first 8, last 8 and 48 pseudo-random rows"""
    import torch

    q, n, batch = Q61, 4096, 4096
    plan = pkg.Plan(q, n)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    pkg.binding.fill_synthetic_dev(q, 0xF4E50002, 0, batch * n, x.data_ptr(), st)
    y = torch.empty_like(x)
    plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)
    z = torch.empty_like(x)
    plan.inverse_dev(y.data_ptr(), z.data_ptr(), batch, st)
    assert torch.equal(z, x)
    rows = list(range(8)) + list(range(batch - 8, batch)) + \
        [int(r) for r in np.random.default_rng(7).integers(8, batch - 8, 48)]
    Y = y.view(batch, n)
    for r in rows:
        a = oracle.fill_synthetic(q, 0xF4E50002, r * n, n)
        assert np.array_equal(Y[r].cpu().numpy().view(np.uint64), oracle.ntt(q, n, a)), f"row {r}"
    assert bool((y.view(torch.int64) >= 0).all()) and int(y.max()) < q   # canonical


def test_config5_n65536_properties(pkg, oracle):
    """configs[4] shape (N=2^16, q61) at a batch that fits one test: round trip, linearity
    NTT(a+b) = NTT(a)+NTT(b) mod q, subset parity against the oracle."""
    import torch

    q, n, batch = Q61, 65536, 512
    plan = pkg.Plan(q, n)
    st = torch.cuda.current_stream().cuda_stream
    dev = "cuda:0"
    a = torch.empty(batch * n, dtype=torch.int64, device=dev)
    b = torch.empty_like(a)
    pkg.binding.fill_synthetic_dev(q, 0xF4E50005, 0, batch * n, a.data_ptr(), st)
    pkg.binding.fill_synthetic_dev(q, 0xF4E50015, 0, batch * n, b.data_ptr(), st)
    A = torch.empty_like(a); B = torch.empty_like(a); S = torch.empty_like(a)
    plan.forward_dev(a.data_ptr(), A.data_ptr(), batch, st)
    plan.forward_dev(b.data_ptr(), B.data_ptr(), batch, st)
    s = a + b
    s = torch.where(s >= q, s - q, s)
    plan.forward_dev(s.data_ptr(), S.data_ptr(), batch, st)
    AB = A + B
    AB = torch.where(AB >= q, AB - q, AB)
    assert torch.equal(S, AB)
    back = torch.empty_like(a)
    plan.inverse_dev(A.data_ptr(), back.data_ptr(), batch, st)
    assert torch.equal(back, a)
    for r in (0, 1, batch // 2, batch - 1):
        row = oracle.fill_synthetic(q, 0xF4E50005, r * n, n)
        assert np.array_equal(A.view(batch, n)[r].cpu().numpy().view(np.uint64), oracle.ntt(q, n, row))


def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _prime_below(limit, step):
    k = (limit - 2) // step
    while k > 0 and not _is_prime(k * step + 1):
        k -= 1
    if k <= 0:
        raise ValueError(f"no prime = 1 mod {step} below {limit}")
    return k * step + 1


@pytest.mark.parametrize("bits", [20, 31, 32, 33, 48, 60, 61, 62])
def test_moduli_across_bit_lengths_and_lazy_range_boundaries(pkg, oracle, bits):
    """The largest NTT-friendly prime below 2^bits for each bit length: 2^61 is where the
    forward kernels switch between the every-other-stage (8q < 2^64) and every-stage lazy
    ranges; 2^32/2^33 exercise the 32-bit limb carries.  Inputs include the extreme patterns
    that maximise the redundant representation."""
    q = _prime_below(1 << bits, 1 << 17)          # q = 1 (mod 2^17): n up to 2^16
    assert (1 << (bits - 1)) < q < (1 << bits)
    for n in (32, 4096, 65536):
        rows = [np.full(n, q - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64)]
        alt = np.zeros(n, dtype=np.uint64); alt[::2] = q - 1; rows.append(alt)
        rows.append(oracle.fill_synthetic(q, bits, 0, n))
        a = np.stack(rows)
        b = np.stack([rows[3], rows[0], rows[0], rows[2]])
        plan = pkg.Plan(q, n)
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.rq_mul(a, b, want_evals=False)[0].reshape(-1),
                              oracle.rq_mul(q, n, a, b)[0].reshape(-1)), (q, n)


@pytest.mark.gpu
def test_host_entry_points_are_reentrant(pkg, oracle):
    """cargo runs the reference's tests on parallel threads, so a shim would enter the library
    concurrently (SURVEY.md §8b "Threading").  Eight host threads mix transforms, products and the
    N1/N2 host wrappers (which share one workspace) at different (q, n); every result is compared
    with the oracle computed up front.  ctypes releases the GIL during the calls."""
    import threading
    B = pkg.binding
    rng = np.random.default_rng(77)
    U64 = 1 << 64
    jobs = []
    for i, (q, n) in enumerate([(Q16, 512), (Q61, 1024), (Q61, 4096), (Q16, 16)]):
        a = rng.integers(0, q, (5 + i, n), dtype=np.uint64)
        b = rng.integers(0, q, (5 + i, n), dtype=np.uint64)
        plan = pkg.Plan(q, n)
        jobs.append((lambda p=plan, x=a: p.forward(x), oracle.ntt(q, n, a)))
        jobs.append((lambda p=plan, x=a, y=b: p.rq_mul(x, y)[0], oracle.rq_mul(q, n, a, b)[0]))
    for n in (64, 1024):
        a = rng.integers(0, U64, (3, n), dtype=np.uint64)
        b = rng.integers(0, U64, (3, n), dtype=np.uint64)
        jobs.append((lambda n=n, x=a, y=b: B.tn_mul(n, x, y), oracle.tn_mul(n, a, b)))
        ia = rng.integers(0, Q16, (4, n), dtype=np.int64)
        ib = rng.integers(0, Q16, (4, n), dtype=np.int64)
        jobs.append((lambda n=n, x=ia, y=ib: B.r_naive_mul(n, x, y), oracle.r_naive_mul(n, ia, ib)))
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for j in range(tid, len(jobs), 3):     # overlapping subsets: same plan on several threads
                    fn, want = jobs[(j + rep) % len(jobs)]
                    got = np.asarray(fn())
                    if not np.array_equal(got.reshape(want.shape), want):
                        errors.append((tid, rep, j))
        except Exception as e:                           # noqa: BLE001 - reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_allocation_failure_is_reported_and_the_library_stays_usable(pkg, oracle):
    """a workspace the device cannot hold (64 TiB) must come back as FHE_E_HIP before anything is
    launched, leave no sticky HIP error behind, and the next call must work"""
    import torch
    B, L = pkg.binding, pkg.load_library()
    n = 65536
    plan = pkg.Plan(Q61, n)
    d = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    rc = L.fhe_rq_mul_dev(plan.handle, d.data_ptr(), 0, d.data_ptr(), 0, d.data_ptr(), None, None, None,
                          1 << 26, None, None)
    assert rc == B.FHE_E_HIP, rc
    assert b"hipMalloc" in L.fhe_last_error() or b"memory" in L.fhe_last_error().lower()
    a = oracle.fill_synthetic(Q61, 9, 0, 2 * n)
    assert np.array_equal(plan.forward(a), oracle.ntt(Q61, n, a))


@pytest.mark.gpu
def test_device_entry_points_can_be_captured_in_a_hip_graph(pkg, oracle):
    """the *_dev entry points only enqueue on the caller's stream (no allocation or synchronisation
    once the plan's tables and the workspace exist), so a launch-bound chain — here an Rq product
    followed by a second one on its result — can be captured once and replayed"""
    import torch
    q, n, batch = Q61, 1024, 3
    plan = pkg.Plan(q, n)
    a = oracle.fill_synthetic(q, 41, 0, batch * n)
    b = oracle.fill_synthetic(q, 42, 0, batch * n)
    as_dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    da, db = as_dev(a), as_dev(b)
    dc, dd = torch.empty_like(da), torch.empty_like(da)
    work = torch.empty(plan.workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")

    def chain(st):
        plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, d_work=work.data_ptr(), stream=st)
        plan.rq_mul_dev(dc.data_ptr(), db.data_ptr(), dd.data_ptr(), batch, d_work=work.data_ptr(), stream=st)

    chain(torch.cuda.current_stream().cuda_stream)      # warm-up: device tables are uploaded here
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(torch.cuda.current_stream().cuda_stream)
    dc.zero_(); dd.zero_()
    g.replay()
    torch.cuda.synchronize()
    c = oracle.rq_mul(q, n, a, b)[0]
    d = oracle.rq_mul(q, n, c, b)[0]
    assert np.array_equal(dc.cpu().numpy().view(np.uint64), c.reshape(-1))
    assert np.array_equal(dd.cpu().numpy().view(np.uint64), d.reshape(-1))
    # new inputs, same graph
    a2 = oracle.fill_synthetic(q, 43, 0, batch * n)
    da.copy_(as_dev(a2))
    g.replay()
    torch.cuda.synchronize()
    c2 = oracle.rq_mul(q, n, a2, b)[0]
    assert np.array_equal(dc.cpu().numpy().view(np.uint64), c2.reshape(-1))


@pytest.mark.gpu
def test_three_kernel_product_path_for_single_pass_sizes(pkg, oracle):
    """For 16 <= n <= 2^13 the product runs as ONE kernel (rq_mul_fused_kernel); FHE_RQ_MUL_FUSED=0
    selects forward + forward + fused-pointwise inverse as for the larger sizes.  Both must give the
    reference's words, with every combination of cached evals and every optional output."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import fhe_study_amd as pkg\n"
        "from oracle import load_oracle\n"
        "O = load_oracle()\n"
        "for q, n, batch in ((65537, 16, 300), (65537, 512, 7), (pkg.Q61, 1024, 5), (pkg.Q61, 8192, 3)):\n"
        "    a = O.fill_synthetic(q, 5, 0, batch * n); b = O.fill_synthetic(q, 6, 0, batch * n)\n"
        "    c, ce, ae, be = O.rq_mul(q, n, a, b)\n"
        "    P = pkg.Plan(q, n)\n"
        "    for got in (P.rq_mul(a, b), P.rq_mul(ae, b, a_is_evals=True), P.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):\n"
        "        assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, (c, ce, ae, be))), (q, n)\n"
        "    assert np.array_equal(P.rq_mul(a, b, want_evals=False)[0].reshape(-1), c.reshape(-1))\n"
        "print('product parity ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for fused in ("0", "1"):
        env = dict(os.environ, FHE_RQ_MUL_FUSED=fused)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "product parity ok" in r.stdout, fused + r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [18, 19, 20])
def test_largest_sizes(pkg, oracle, log_n):
    """n = 2^18 .. 2^20 (the engine's limit): 8 strided + 10..12 contiguous stages, the only sizes
    whose contiguous pass is longer than 8 stages AND not the whole transform"""
    q, n, batch = Q61, 1 << log_n, 2
    a = oracle.fill_synthetic(q, 300 + log_n, 0, batch * n)
    b = oracle.fill_synthetic(q, 400 + log_n, 0, batch * n)
    a[:4] = [0, q - 1, 1, q - 1]
    plan = pkg.Plan(q, n)
    A = plan.forward(a)
    assert np.array_equal(A, oracle.ntt(q, n, a))
    assert np.array_equal(plan.inverse(A), a)
    assert np.array_equal(plan.inverse(b), oracle.intt(q, n, b))
    c, ce, ae, be = plan.rq_mul(a, b)
    oc, oce, oae, obe = oracle.rq_mul(q, n, a, b)
    assert np.array_equal(c.reshape(-1), oc.reshape(-1)) and np.array_equal(ce.reshape(-1), oce.reshape(-1))
