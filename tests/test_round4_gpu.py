"""Round 4: the one-launch n = 2^16 forward transform (fhe-study_amd/csrc/ntt_persist.hip) — persistent workgroups, one
ticket queue per XCD, in two forms: "A" (tiles, the strided stages lagging ahead of the contiguous ones) and "B" (teams of
sixteen workgroups taking one polynomial through both halves, the intermediate read back out of the L2; "D": the same at
two workgroups per CU with 256 registers and the next part prefetched; "E": the teams without the meeting — a workgroup
keeps a FIFO of the parts whose contiguous half is still to come and runs whichever half is ready).  NTT::ntt is
arith/src/ntt.rs:44-73; the words must be the two-pass kernels' and the oracle's, whatever the settings, the batch shape
and the number of workgroups the launch gets."""
import numpy as np
import pytest

from conftest import Q61

N = 1 << 16
SETTINGS = [("A", 1, 4, 6), ("A", 4, 2, 4), ("A", 16, 1, 0), ("A", 16, 1, 3), ("A", 64, 0, 0), ("A", 256, 1, 0),
            ("B", 1, 0, 1), ("B", 1, 1, 2), ("B", 1, 2, 4), ("D", 1, 1, 1), ("D", 1, 0, 2), ("E", 1, 1, 2), ("E", 1, 0, 3),
            ("E", 1, 1, 6)]


@pytest.fixture()
def need_gpu(pkg):
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"


@pytest.fixture()
def persist_off(pkg):
    yield
    pkg.binding.set_persist(0)
    pkg.binding.set_persist_grid(0)


def _rows(oracle, q, n, batch, seed):
    a = oracle.fill_synthetic(q, seed, 0, batch * n).reshape(batch, n)
    a[0] = 0
    if batch > 1:
        a[1] = q - 1
    if batch > 2:
        a[2, ::2] = q - 1
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: f"{s[0]}:{s[1]},{s[2]},{s[3]}")
def test_persistent_forward_is_the_oracles_transform(pkg, oracle, need_gpu, persist_off, setting):
    """every setting, batches that leave tiles, items, teams and queues ragged; the first rows against the oracle, all rows
    against the two-pass kernels; extreme inputs (all zero, all q - 1, alternating) among them"""
    plan = pkg.Plan(Q61, N)
    for batch in (1, 2, 9, 37, 130):
        a = _rows(oracle, Q61, N, batch, 4000 + batch)
        pkg.binding.set_persist(0)
        ref = plan.forward(a)
        pkg.binding.set_persist(*setting)
        got = plan.forward(a)
        assert np.array_equal(got, ref), (setting, batch)
        k = min(batch, 4)
        assert np.array_equal(got[:k].reshape(-1), oracle.ntt(Q61, N, a[:k]).reshape(-1)), (setting, batch)
        assert np.array_equal(plan.inverse(got), a), (setting, batch)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [1, 3, 8, 9, 20, 100, 700])
def test_no_workgroup_needs_another_to_be_resident(pkg, oracle, need_gpu, persist_off, grid):
    """The same words from ANY number of workgroups: "A" from a single one upwards; "B" whenever every XCD runs at least
    one (with fewer than eight workgroups an XCD's queue is never served — that must come back as an error, FHE_E_HIP, not
    as a hole in the output; from eight upwards the round-robin placement gives every XCD one, a lone workgroup running all
    sixteen parts of its polynomials by itself)."""
    plan = pkg.Plan(Q61, N)
    batch = 19
    a = _rows(oracle, Q61, N, batch, 4100)
    pkg.binding.set_persist(0)
    ref = plan.forward(a)
    pkg.binding.set_persist_grid(grid)
    for setting in (("A", 4, 1, 3), ("A", 16, 1, 0), ("B", 1, 1, 2), ("D", 1, 1, 1), ("E", 1, 1, 2), ("E", 1, 1, 4)):
        pkg.binding.set_persist(*setting)
        if setting[0] in "BDE" and grid < 8:
            with pytest.raises(Exception) as e:     # (FheError: the package may be loaded under two module names in one session)
                plan.forward(a)
            assert type(e.value).__name__ == "FheError" and e.value.code == pkg.binding.FHE_E_HIP and "never served" in str(e.value)
            continue
        assert np.array_equal(plan.forward(a), ref), (setting, grid)


@pytest.mark.gpu
def test_persistent_forward_on_device_buffers_in_place_and_repeated(pkg, oracle, need_gpu, persist_off):
    """the *_dev entry point: out of place, in place, and the same launch repeated on one stream (the control block and
    the ring are reused from call to call), 300 polynomials"""
    import torch

    B = pkg.binding
    plan = pkg.Plan(Q61, N)
    st = torch.cuda.current_stream().cuda_stream
    batch = 300
    x = torch.empty(batch * N, dtype=torch.int64, device="cuda:0")
    B.fill_synthetic_dev(Q61, 41, 0, batch * N, x.data_ptr(), st)
    ref = torch.empty_like(x)
    B.set_persist(0)
    plan.forward_dev(x.data_ptr(), ref.data_ptr(), batch, st)
    torch.cuda.synchronize()
    for setting in (("A", 16, 1, 0), ("A", 2, 3, 5), ("B", 1, 1, 2), ("D", 1, 1, 2), ("E", 1, 1, 3)):
        B.set_persist(*setting)
        y = torch.zeros_like(x)
        for _ in range(3):
            plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)
        z = x.clone()
        plan.forward_dev(z.data_ptr(), z.data_ptr(), batch, st)
        torch.cuda.synchronize()
        B.persist_status()
        assert torch.equal(y, ref) and torch.equal(z, ref), setting


@pytest.mark.gpu
def test_other_plans_keep_the_two_pass_kernels(pkg, oracle, need_gpu, persist_off):
    """the persistent kernels exist for n = 2^16 on a pseudo-Mersenne modulus; every other plan ignores the setting"""
    pkg.binding.set_persist("B", 1, 1, 2)
    for q, n in ((Q61, 1 << 15), (Q61, 4096), (0x1ffffffffc000001, 1 << 16), (65537, 1 << 15)):
        if (q - 1) % (2 * n):
            continue
        a = _rows(oracle, q, n, 3, 4200 + n)
        assert np.array_equal(pkg.Plan(q, n).forward(a).reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)


def test_set_persist_rejects_what_the_kernels_cannot_run(pkg):
    B = pkg.binding
    for bad in ((1, 3, 1, 0), (1, 2048, 1, 0), (1, 16, 2, 2), (2, 1, 0, 0), (3, 1, 1, 0), (4, 1, 1, 1), (5, 1, 1, 2)):
        with pytest.raises(Exception) as e:
            B.set_persist(*bad)
        assert type(e.value).__name__ == "FheError" and e.value.code == B.FHE_E_INVALID
    B.set_persist(0)


@pytest.mark.gpu
def test_two_threads_on_the_default_stream(pkg, oracle, need_gpu):
    """Two host threads enqueue products that need the library workspace (two-pass size, d_work = NULL) on the SAME stream —
    the NULL default stream — with no synchronisation between them: the workspace is keyed by the calling thread as well,
    so their kernel sequences may interleave on the stream without sharing an intermediate (until round 4 the
    header forbade this and two such threads silently computed wrong words)."""
    import threading

    import torch

    q, n, batch = Q61, 16384, 12
    plan = pkg.Plan(q, n)
    jobs = []
    for i in range(2):
        a = oracle.fill_synthetic(q, 700 + i, 0, batch * n)
        b = oracle.fill_synthetic(q, 800 + i, 0, batch * n)
        da = torch.from_numpy(a.view(np.int64).copy()).cuda()
        db = torch.from_numpy(b.view(np.int64).copy()).cuda()
        jobs.append((da, db, torch.empty_like(da), oracle.rq_mul(q, n, a, b)[0]))
    torch.cuda.synchronize()
    errs = []

    def work(j):
        da, db, dc, _ = jobs[j]
        try:
            for _ in range(40):
                plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, stream=None)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for _, _, dc, want in jobs:
        assert np.array_equal(dc.cpu().numpy().view(np.uint64), want)


# ---- q = 1 (mod 2^32): the forward transforms on the word-Montgomery table (zq_device.hpp: ct_bfly_mg) ----------------------
QMG61, QMG50, QMG40 = 0x1ffffff900000001, 0x3fff300000001, 0xff00000001


@pytest.mark.gpu
@pytest.mark.parametrize("q,n", [(QMG61, 16), (QMG61, 64), (QMG61, 256), (QMG61, 2048), (QMG61, 8192), (QMG61, 1 << 14),
                                 (QMG61, 1 << 15), (QMG61, 1 << 16), (QMG61, 1 << 17), (QMG50, 32), (QMG50, 4096),
                                 (QMG50, 1 << 16), (QMG40, 16), (QMG40, 1024), (QMG40, 1 << 14)])
def test_montgomery_forward_is_the_oracles_transform(pkg, oracle, need_gpu, q, n):
    """NTT::ntt (arith/src/ntt.rs:44-73) for q = qh 2^32 + 1 below 2^61: the forward kernels run the word-Montgomery
    butterflies (plan arithmetic 5) in both directions — the words are the oracle's, on zeros, on q - 1 everywhere (the
    largest values every bound has to hold for) and on ragged batches."""
    import os

    plan = pkg.Plan(q, n)
    assert plan.arithmetic() == (5 if os.environ.get("FHE_MG", "1")[:1] != "0" else 1)
    batch = 5 if n >= (1 << 14) else 37
    a = _rows(oracle, q, n, batch, 4300 + n)
    A = plan.forward(a)
    assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (hex(q), n)
    assert np.array_equal(plan.inverse(A).reshape(-1), a.reshape(-1)), (hex(q), n)
    # NTT::intt (ntt.rs:78-110) on the same rows taken as evaluations — zeros and q - 1 everywhere among them
    assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (hex(q), n)


@pytest.mark.gpu
@pytest.mark.parametrize("q,n", [(QMG61, 16), (QMG61, 128), (QMG61, 1024), (QMG61, 8192), (QMG61, 1 << 14), (QMG61, 1 << 16),
                                 (QMG50, 64), (QMG50, 4096), (QMG50, 1 << 15), (QMG40, 16), (QMG40, 2048), (QMG40, 1 << 14)])
def test_montgomery_products_are_the_oracles(pkg, oracle, need_gpu, q, n):
    """Rq x Rq (ring_nq.rs:586-607) on such a modulus — fused at single-pass sizes, strided | middle | strided above — with
    the variable x variable product on the Montgomery constants (zq_device.hpp: mul_var_mg): c, c_evals, a_evals, b_evals are
    the oracle's words, on q - 1 everywhere and zeros too, and with operands handed over as evaluations."""
    plan = pkg.Plan(q, n)
    batch = 3 if n >= (1 << 14) else 19
    a, b = _rows(oracle, q, n, batch, 4400 + n), _rows(oracle, q, n, batch, 4500 + n)
    b[0] = q - 1                                                # (q - 1) x 0, (q - 1) x (q - 1), ...
    got = plan.rq_mul(a, b)
    want = oracle.rq_mul(q, n, a, b)
    for g, w, name in zip(got, want, ("c", "c_evals", "a_evals", "b_evals")):
        assert np.array_equal(np.asarray(g).reshape(-1), np.asarray(w).reshape(-1)), (hex(q), n, name)
    A, B_ = want[2], want[3]
    for flags, x, y in (((True, False), A, b), ((False, True), a, B_), ((True, True), A, B_)):
        c = plan.rq_mul(x, y, a_is_evals=flags[0], b_is_evals=flags[1], want_evals=False)[0]
        assert np.array_equal(np.asarray(c).reshape(-1), np.asarray(want[0]).reshape(-1)), (hex(q), n, flags)


@pytest.mark.gpu
def test_montgomery_plans_on_device_buffers_in_place(pkg, oracle, need_gpu):
    """the in-place device entry point on such a modulus"""
    import torch

    n, batch = 1 << 16, 40
    plan = pkg.Plan(QMG61, n)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    pkg.binding.fill_synthetic_dev(QMG61, 77, 0, batch * n, x.data_ptr(), st)
    host = x.cpu().numpy().view(np.uint64).reshape(batch, n)
    plan.forward_dev(x.data_ptr(), x.data_ptr(), batch, st)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy().view(np.uint64).reshape(-1), oracle.ntt(QMG61, n, host).reshape(-1))


def _mul_mg(y, wa, wb, q, p):
    """the nine instructions of FHE_MG_PRODUCT (zq_device.hpp) in Python integers, word for word"""
    M32, M64 = (1 << 32) - 1, (1 << 64) - 1
    y0, y1 = y & M32, y >> 32
    a0, a1, b0, b1 = wa & M32, wa >> 32, wb & M32, wb >> 32
    N = y0 * a0
    N = N + y1 * b0
    carry, N = N >> 64, N & M64                       # v_mad_u64_u32 ... carry -> vcc
    B = (N >> 32) | (carry << 32)                     # v_mov / v_addc: {n1, carry}
    B = B + y0 * a1
    B = B + y1 * b1
    assert B <= M64 and B < 2 * q                     # T >> 32 < 2q
    T0 = N & M32
    R = (B + T0 * p["nqh"]) & M64                     # v_mad_u64_u32 (wraps)
    R = ((((R >> 32) - T0) & M32) << 32) | (R & M32)  # v_sub_u32 on the high word
    return (R + q) & M64                              # v_lshl_add_u64


def test_montgomery_product_identity_and_bounds():
    """r = y w (mod q) and 0 < r < 3q for ANY 64-bit y on the table {w 2^32, w 2^64 mod q}; the forward schedule of
    ntt_rounds.hpp (AK = 4: a conditional subtraction of 4q before every stage from the third on) keeps every value below 2^64"""
    import random
    from fhe_study_amd.arith import mg_params, pm_params

    rng = random.Random(5)
    for q in (QMG61, QMG50, QMG40):
        p = mg_params(q)
        assert p is not None and pm_params(q) is None
        for _ in range(4000):
            w = rng.randrange(q)
            y = rng.choice([rng.getrandbits(64), (1 << 64) - 1, q - 1, 0, 1, 8 * q - 1 if 8 * q < 1 << 64 else q])
            wa = (w << 32) % q
            wb = (wa << 32) % q
            r = _mul_mg(y, wa, wb, q, p)
            assert 0 < r < 3 * q and r % q == (y * w) % q
        # sixteen stages on the worst values the bounds allow: x' = u + r, y' = u + 3q - r, u reduced when its bound asks
        bx = 16                                             # sixteenths of q
        x = q - 1
        for stage in range(16):
            if bx + 48 > 128:
                assert x < 8 * q
                x = x - 4 * q if x >= 4 * q else x
                bx = 64
            r = 3 * q - 1
            x, yv = x + r, x + 3 * q - 1
            bx += 48
            assert x < (1 << 64) and yv < (1 << 64) and x * 16 < bx * q + 16
        assert bx == 112
    for q in (Q61, 0x1fffffffff000001, 65537, (1 << 32) + 1 + (1 << 61)):
        assert mg_params(q) is None


def test_montgomery_variable_product_identity_and_bounds():
    """mul_var_mg's two word steps on the full 128-bit product, in Python integers: (P - W q) / 2^32 is exact twice over because
    q = 1 (mod 2^32); the result + q is a b 2^-64 (mod q), positive and below 7.1 q for every a, b below 7q"""
    import random

    rng = random.Random(9)
    for q in (QMG61, QMG50, QMG40):
        qh, inv64 = q >> 32, pow(1 << 64, -1, q)
        for _ in range(20000):
            a = rng.choice([rng.randrange(7 * q), 7 * q - 1, q - 1, 0, 1])
            b = rng.choice([rng.randrange(7 * q), 7 * q - 1, q - 1, 0, 1])
            P = a * b
            assert P < 1 << 128
            P0 = P & 0xffffffff
            assert (P - P0 * q) % (1 << 32) == 0
            R1 = (P >> 32) - P0 * qh
            assert R1 == (P - P0 * q) >> 32
            R10 = R1 & 0xffffffff
            R2 = (R1 >> 32) - R10 * qh
            r = R2 + q
            assert 0 < r < 7.1 * q and r < 1 << 64 and r % q == (a * b * inv64) % q


# ---- 32-bit inverse rounds without a conditional subtraction per butterfly (ntt32_rounds.hpp: inv32_sched / round_inv32_loose) ---------
def _inv32_sched(R, bin_, bout, cap=16):
    """inv32_sched restated: per stage and butterfly (reduce x, reduce y, K), and which registers are reduced at the end"""
    B, stages = [bin_] * 16, []
    for i in range(R - 1, -1, -1):
        span, st = 8 >> i, []
        for g in range(1 << i):
            for l in range(span):
                k, k2 = g * 2 * span + l, g * 2 * span + l + span
                bx, by, rx, ry = B[k], B[k2], False, False
                if bx + by > cap:
                    if bx >= by:
                        rx, bx = True, 2
                    else:
                        ry, by = True, 2
                if bx + by > cap:
                    if not rx:
                        rx, bx = True, 2
                    else:
                        ry, by = True, 2
                st.append((k, k2, rx, ry, by))
                B[k], B[k2] = bx + by, 2
        stages.append(st)
    return stages, [b > bout for b in B], B


def test_loose_inverse_rounds_stay_in_a_word_and_compute_the_same_residues():
    """For the 27-bit primes of digit32.hpp (p < 2^32 / 25): every sum and every x - y + K p of the schedule stays below 2^32 and
    above -1 for the LARGEST values its bounds allow, a round hands on values below its BOUT, and on random words the registers
    are congruent to what the plain Gentleman-Sande stages give — for every (R, BIN, BOUT) the block kernels instantiate."""
    import random

    rng = random.Random(11)
    for p in (0x0a3c8001, 0x0a320001, 0x0a318001):                 # kExt32PrimeA / B / C (digit32.hpp): all below 2^32 / 25
        assert p < (1 << 32) // 25
        for R, bin_, bout in ((4, 2, 4), (4, 4, 4), (1, 4, 2), (1, 4, 8), (2, 4, 2), (3, 4, 8)):
            stages, fin, B = _inv32_sched(R, bin_, bout)
            for b, f in zip(B, fin):
                assert (2 if f else b) <= bout
            # worst case: every register at its bound - 1
            bound = [bin_] * 16
            for st in stages:
                for k, k2, rx, ry, K in st:
                    bx, by = (2 if rx else bound[k]), (2 if ry else bound[k2])
                    assert K >= by and bx + by <= 16 and (bx + K) * p < 1 << 32 and (bx + by) * p < 1 << 32
                    bound[k], bound[k2] = bx + by, 2
            # random words: congruence with the exact butterflies
            w = [rng.randrange(1, p) for _ in range(16)]
            v = [rng.randrange(bin_ * p) for _ in range(16)]
            exact = list(v)
            for si, st in enumerate(stages):
                for k, k2, rx, ry, K in st:
                    x, y = v[k], v[k2]
                    x = x % p + (p if rx and rng.random() < 0.5 else 0) if rx else x     # Barrett lands in [0, 2p)
                    y = y % p + (p if ry and rng.random() < 0.5 else 0) if ry else y
                    d = x - y + K * p
                    assert 0 <= d < 1 << 32 and x + y < 1 << 32
                    v[k], v[k2] = x + y, (d * w[si]) % p + (p if rng.random() < 0.5 else 0)   # the lazy product lands in [0, 2p)
                    ex, ey = exact[k], exact[k2]
                    exact[k], exact[k2] = (ex + ey) % p, ((ex - ey) * w[si]) % p
            for k in range(16):
                assert v[k] % p == exact[k] and (v[k] < B[k] * p)
