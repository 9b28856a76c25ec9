"""Round 3: the five-multiply butterflies for pseudo-Mersenne moduli (csrc/zq_device.hpp, `AR == 2` kernels).

q = 2^k - delta with 56 <= k <= 61 and delta <= 2^(k-39) — SURVEY.md section 8's modulus 2^61 - 2^21 + 1 is one — runs
`NTT::ntt` / `intt` / `Rq x Rq` (arith/src/ntt.rs:44-110, ring_nq.rs:586-607) on a second pair of tables
{w, w 2^32 mod q}.  The CPU part restates that arithmetic in Python integers and checks the identities and the value
bounds the kernels rely on; the GPU part is word-for-word parity with the oracle at every such modulus shape, and
identity with the Shoup kernels (FHE_PM=0) on the same inputs.
"""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import Q61, ROOT

# NTT-friendly primes (q = 1 mod 2^17) of the pseudo-Mersenne form, and two that are NOT eligible
PM_PRIMES = [
    (2305843009211596801, 61),   # 2^61 - 2^21 + 1      (delta = 2^21 - 1)
    (2305843009210023937, 61),   # 2^61 - 28 * 2^17 + 1 (delta just under 2^22, the limit for k = 61)
    (1152921504606584833, 60),   # 2^60 - 2^18 + 1
    (144115188075593729, 57),    # 2^57 - 2^18 + 1      (delta = 2^18 - 1 <= 2^(57-39))
]
NOT_PM = [
    2305843009208713217,         # 2^61 - 38 * 2^17 + 1: delta > 2^22
    576460752300015617,          # 2^59 - 26 * 2^17 + 1: delta > 2^20
]


def test_eligibility_rule_matches_the_library_statement():
    from fhe_study_amd.arith import pm_params

    for q, k in PM_PRIMES:
        p = pm_params(q)
        assert p is not None and p["k"] == k and p["delta"] == (1 << k) - q
        assert p["delta"] <= 1 << (k - 39)
    for q in NOT_PM + [65537, 12289, 4611686018425815041, 0x1fffffffff000001]:
        assert pm_params(q) is None


def _mul_pm(y, w, w2, p):
    """the nine instructions of FHE_PM_PRODUCT (zq_device.hpp) in Python integers, word for word"""
    M32, M64 = (1 << 32) - 1, (1 << 64) - 1
    y0, y1 = y & M32, y >> 32
    a0, a1, b0, b1 = w & M32, w >> 32, w2 & M32, w2 >> 32
    N = y0 * a0
    N = N + y1 * b0
    carry, N = N >> 64, N & M64                       # v_mad_u64_u32 ... carry -> vcc
    B = (N >> 32) | (carry << 32)                     # v_mov / v_addc: {n1, carry}
    B = B + y0 * a1
    B = B + y1 * b1
    assert B <= M64                                   # T >> 32 fits 64 bits
    lo = (N & M32) | ((B & M32 & p["mask"]) << 32)    # T mod 2^(k+1)
    th = (B >> p["sh"])
    assert th <= M32                                  # T >> (k+1) fits 32 bits (v_alignbit takes the low word)
    r = th * p["c2"] + lo
    assert r <= M64
    return r


def test_five_multiply_product_identity_and_bounds():
    """r = y w (mod q) and r < 2q + q/16 for ANY 64-bit y; the reduction lands below q + q/16; a butterfly's outputs
    stay below 2^64 under the schedule of ntt_rounds.hpp (bounds in sixteenths of q)."""
    from fhe_study_amd.arith import pm_params

    rnd = random.Random(0xF4E5_0300)
    for q, k in PM_PRIMES:
        p = pm_params(q)
        ys = [0, 1, (1 << 64) - 1, (1 << 63), q - 1, q, 8 * q - 1 if 8 * q - 1 < (1 << 64) else (1 << 64) - 1,
              (1 << 32) - 1, 1 << 32, ((1 << 32) - 1) << 32]
        ws = [0, 1, q - 1, q // 2, (1 << 32) - 1, 1 << 32]
        for _ in range(300):
            ys.append(rnd.getrandbits(64))
            ws.append(rnd.randrange(q))
        for y in ys:
            for w in ws[:6] + rnd.sample(ws[6:], 12):
                w2 = (w << 32) % q
                r = _mul_pm(y, w, w2, p)
                assert r % q == (y * w) % q
                assert 16 * r < 33 * q                 # kPmMul
            # pm_reduce
            x1 = y >> 32
            red = (x1 >> p["rsh"]) * p["delta"] + (((x1 & p["rmask"]) << 32) | (y & 0xFFFFFFFF))
            assert red % q == y % q and 16 * red < 17 * q      # kPmRed
        # forward stage: x < 5q (80 sixteenths) needs no reduction: x' = u + r, y' = u - r + 3q stay below 2^64
        assert 5 * q + (33 * q) // 16 + 1 < 1 << 64 and 5 * q + 3 * q <= 1 << 64
        # inverse stage: d = x - y + K q with K q >= bound of y: the largest pair the schedule admits
        assert 8 * q <= 1 << 64


def test_variable_times_variable_on_the_same_product():
    """zip_eq(l,r).map(l*r) (ring_nq.rs:601-604) with the multiplier's second table word formed on the fly
    (zq_device.hpp: pm_shift32): exact for every multiplicand below 7.9 q and every multiplier below 2^k, and the
    double reduction lands strictly below 2^k"""
    from fhe_study_amd.arith import pm_params

    rnd = random.Random(0xF4E5_0301)
    for q, k in PM_PRIMES:
        p = pm_params(q)

        def red(x):
            x1 = x >> 32
            return (x1 >> p["rsh"]) * p["delta"] + (((x1 & p["rmask"]) << 32) | (x & 0xFFFFFFFF))

        amax = min((1 << 64) - 1, (127 * q) // 16)
        As = [0, 1, q - 1, q, amax, (113 * q) // 16] + [rnd.randrange(amax + 1) for _ in range(200)]
        Bs = [0, 1, q - 1, q, (1 << k) - 1, (1 << (k - 32)) - 1, 1 << (k - 32)] + [rnd.randrange(1 << k) for _ in range(200)]
        for a in As:
            for b in Bs[:7] + rnd.sample(Bs[7:], 10):
                wh = ((b >> 32 << 32 | (b & 0xFFFFFFFF)) >> p["rsh"]) & 0xFFFFFFFF
                assert wh == b >> p["rsh"]                                   # fits 32 bits for b < 2^k
                w2 = wh * p["delta"] + (((b & 0xFFFFFFFF) & p["rmask"]) << 32)
                assert w2 % q == (b << 32) % q and w2 < (1 << k) + (1 << (k - 7))
                r = _mul_pm(a, b, w2, p)                                      # asserts T >> (k+1) < 2^32 inside
                assert r % q == (a * b) % q and 16 * r < 33 * q
        for x in [0, (1 << 64) - 1, 8 * q - 1 if 8 * q <= 1 << 64 else (1 << 64) - 1, (1 << k) - 1, 1 << k, (1 << k) + 7 * p["delta"]] + \
                 [rnd.getrandbits(64) for _ in range(500)]:
            y = red(red(x))
            assert y % q == x % q and y < 1 << k


def test_round_schedules_never_pass_the_cap():
    """the compile-time schedules of ntt_rounds.hpp restated: forward bounds per stage, inverse per register"""
    CAP, MUL, RED, ONE = 128, 33, 17, 16

    def fwd_out(R, b):
        for _ in range(R):
            b = (RED if b + 3 * ONE > CAP else b) + 3 * ONE
            assert b <= CAP
        return b

    assert fwd_out(8, ONE) == 113 and fwd_out(8, 113) == 113        # kPmPassBound, for every split of 8 stages into rounds
    for r0 in (2, 3, 4):
        assert fwd_out(4, fwd_out(r0, ONE)) <= 113                   # strided passes of 6 / 7 / 8 stages
    for lp in range(4, 14):                                          # contiguous passes from either input bound
        assert fwd_out(lp, 113) <= 113 and fwd_out(lp, ONE) <= 113

    def fits(bx, by):
        return bx + by <= CAP and bx + ONE * ((by + ONE - 1) // ONE) <= CAP

    def inv_round(R, bin_, fold):
        B = [bin_] * 16
        for i in range(R - 1, -1, -1):
            span = 8 >> i
            for g in range(1 << i):
                for l in range(span):
                    k = g * 2 * span + l
                    k2 = k + span
                    bx, by = B[k], B[k2]
                    rx = ry = False
                    if not fits(bx, by):
                        if bx >= by:
                            rx, bx = True, RED
                        else:
                            ry, by = True, RED
                    if not fits(bx, by):
                        if not rx:
                            bx = RED
                        else:
                            by = RED
                    assert fits(bx, by)
                    B[k] = MUL if (fold and i == 0) else bx + by
                    B[k2] = MUL
        return [min(b, MUL) if b > 33 else b for b in B]              # the final reductions bring every register below 33

    for R in (1, 2, 3, 4):
        for bin_ in (ONE, 33):
            for fold in (False, True):
                assert max(inv_round(R, bin_, fold)) <= 33


# ---- GPU --------------------------------------------------------------------------------------------------------------


@pytest.fixture()
def need_gpu(pkg):
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"


def _extreme_rows(oracle, q, n, seed):
    rows = [np.zeros(n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64)]
    alt = np.zeros(n, dtype=np.uint64); alt[::2] = q - 1; rows.append(alt)
    e = np.zeros(n, dtype=np.uint64); e[n - 1] = q - 1; rows.append(e)
    rows.append(oracle.fill_synthetic(q, seed, 0, n))
    rows.append(oracle.fill_synthetic(q, seed + 1, 0, n))
    return np.stack(rows)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [p[0] for p in PM_PRIMES] + NOT_PM)
def test_pseudo_mersenne_moduli_every_kernel_shape(pkg, oracle, need_gpu, q):
    """tiny (n < 16: Shoup tables), every single-pass round structure (R0 = 1..4, 1..4 rounds), the two-pass sizes
    with 6, 7 and 8 strided stages and a contiguous pass longer than 8 stages; forward, inverse, product with every
    evals combination, on inputs that maximise the lazy representation"""
    for n in (4, 16, 32, 128, 512, 2048, 8192, 1 << 14, 1 << 15, 1 << 16):
        if (q - 1) % (2 * n):
            continue
        a = _extreme_rows(oracle, q, n, 3000 + n)
        b = a[::-1].copy()
        plan = pkg.Plan(q, n)
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(A).reshape(-1), a.reshape(-1)), (q, n)
        want = oracle.rq_mul(q, n, a, b)
        ae, be = want[2], want[3]
        for got in (plan.rq_mul(a, b), plan.rq_mul(ae, b, a_is_evals=True), plan.rq_mul(a, be, b_is_evals=True),
                    plan.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):
            assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), (q, n)


@pytest.mark.gpu
def test_largest_size_on_the_pseudo_mersenne_tables(pkg, oracle, need_gpu):
    q, n = Q61, 1 << 18            # 8 strided + 10 contiguous stages
    a = _extreme_rows(oracle, q, n, 77)[1:5]
    plan = pkg.Plan(q, n)
    A = plan.forward(a)
    assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1))
    assert np.array_equal(plan.inverse(A).reshape(-1), a.reshape(-1))


@pytest.mark.gpu
def test_shoup_and_pseudo_mersenne_kernels_give_the_same_words(pkg, oracle, need_gpu):
    """FHE_PM=0 keeps the modulus on the ten-multiply Shoup kernels: same words as the default build, and as the
    oracle, for the transform, its inverse and the product — at the bench's shape too (N = 2^16)."""
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "import fhe_study_amd as pkg\n"
        "from oracle import load_oracle\n"
        "O = load_oracle()\n"
        "q = pkg.Q61\n"
        "for n, batch in ((64, 9), (4096, 5), (16384, 3), (65536, 18)):\n"
        "    a = O.fill_synthetic(q, 9, 0, batch * n); b = O.fill_synthetic(q, 10, 0, batch * n)\n"
        "    P = pkg.Plan(q, n)\n"
        "    A = P.forward(a); I = P.inverse(b); c, ce, ae, be = P.rq_mul(a, b)\n"
        "    if n < 65536 or True:\n"
        "        assert np.array_equal(A, O.ntt(q, n, a)) and np.array_equal(I, O.intt(q, n, b))\n"
        "    h = hashlib.sha256()\n"
        "    for x in (A, I, c, ce, ae, be): h.update(np.ascontiguousarray(x).tobytes())\n"
        "    print('digest', n, h.hexdigest())\n" % ROOT)
    outs = {}
    for pm in ("0", "1"):
        env = dict(os.environ, FHE_PM=pm)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, pm + r.stdout + r.stderr
        outs[pm] = [l for l in r.stdout.splitlines() if l.startswith("digest")]
        assert len(outs[pm]) == 4
    assert outs["0"] == outs["1"]


# ---- multi-GPU rehearsals on the one GPU of the box ------------------------------------------------


def _run_bench(args, nproc, timeout=900):
    from test_sharding_gloo import _free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_through_rccl_at_world_size_one(pkg, need_gpu):
    """bench.py as ONE rank under the launcher with --backend nccl --force-dist: init_process_group("nccl"), the
    barriers, the MAX all-reduce of the elapsed time on a DEVICE tensor and all_gather_into_tensor of the rank's real
    shard all execute in librccl — nothing of the N > 1 line is first run on the day an 8-GPU node appears."""
    out = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--backend", "nccl", "--force-dist", "--allgather",
                      "--batch-per-gpu", "256", "--no-cpu-baseline", "--parity-all-ranks"], 1)
    assert out["n_gpus"] == 1 and out["distributed"]["backend"].startswith("rccl") and out["distributed"]["forced_at_world_1"]
    assert out["parity"]["mismatching_rows"] == 0 and out["parity"]["ranks_checked"] == 1
    ag = out["allgather"]
    assert ag["backend"] == "rccl" and ag["rows_per_rank"] == 256 and ag["rows_gathered"] == 256
    assert ag["bytes_sent_per_rank"] == 256 * 65536 * 8 and ag["own_rows_in_place"] is True and ag["seconds"] > 0
    # chunked form of the same collective
    out = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--backend", "nccl", "--force-dist", "--allgather",
                      "--allgather-chunk-rows", "96", "--global-batch", "256", "--no-cpu-baseline", "--no-parity"], 1)
    assert out["allgather"]["collectives"] == 3 and out["allgather"]["own_rows_in_place"] is True
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 256


@pytest.mark.gpu
def test_bench_strong_scaling_two_ranks_on_one_gpu(pkg, need_gpu):
    """BASELINE.json configs[4] fixes the GLOBAL batch: two ranks (sharing cuda:0, gloo rendezvous) split 512 + 1
    polynomials by fhe_shard_range (257 + 256), both shards against the oracle, shards gathered == single-rank
    transform, and the all-gather of the real shards reported on its own."""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu", "--backend", "gloo",
                      "--global-batch", "513", "--no-cpu-baseline", "--parity-all-ranks", "--gather-check", "--allgather"], 2)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 513 and out["config"]["batch_per_gpu"] == 257
    assert "strong scaling" in out["config"]["workload"]
    assert out["parity"]["mismatching_rows"] == 0 and out["parity"]["ranks_checked"] == 2
    assert out["gather_check"]["rows"] == 513 and out["gather_check"]["equal_to_single_rank_transform"] is True
    assert out["gather_check"]["overlapped_equal"] is True and out["gather_check"]["overlapped_pieces_of_rows"] == 64
    assert out["allgather"]["rows_per_rank"] == 257 and out["allgather"]["rows_gathered"] == 513
    # value counts the GLOBAL batch once per step
    assert abs(out["value"] - 513 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6


@pytest.mark.gpu
def test_default_line_is_the_headline_configuration(pkg, need_gpu):
    """no flags beyond the driver's: global batch 65536 on one GPU, the modulus of SURVEY.md section 8, the
    five-multiply arithmetic — checked on a short run (2 steps)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 65536 and out["config"]["batch_per_gpu"] == 65536
    want = "pseudo-mersenne" if os.environ.get("FHE_PM", "1")[:1] != "0" else "shoup61"      # FHE_PM=0: the A/B switch
    assert out["config"]["q"] == Q61 and out["config"]["n"] == 65536 and out["config"]["arithmetic"] == want
    assert out["scaling"] == "strong" and out["parity"]["mismatching_rows"] == 0
    assert out["roofline"]["step_frac"] > 0.2


# ---- advisor findings of round 2 ------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("q,sizes", [(0x0a3c8001, (4096, 8192, 16384)), (0x0a320001, (1024, 65536))])
def test_small_modulus_kernels_at_the_top_of_their_range(pkg, oracle, need_gpu, q, sizes):
    """smallq.hip admits q with 25 q < 2^32; its value bounds (25 q after twelve loose stages, the reduction points
    of the big transforms, 4 q^2 < q 2^32 in the Montgomery product) are tight only at the top of that range:
    25 * 0x0a3c8001 = 2^32 - 1.5 M.  Forward, inverse and the product with every evals combination, on inputs that
    maximise the redundant representation."""
    assert 25 * q < 1 << 32
    for n in sizes:
        assert (q - 1) % (2 * n) == 0
        a = _extreme_rows(oracle, q, n, 4000 + n)
        b = a[::-1].copy()
        plan = pkg.Plan(q, n)
        assert plan.arithmetic() == (3 if os.environ.get("FHE_EXT32", "1")[:1] != "0" else 1)
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n)
        want = oracle.rq_mul(q, n, a, b)
        ae, be = want[2], want[3]
        for got in (plan.rq_mul(a, b), plan.rq_mul(ae, b, a_is_evals=True), plan.rq_mul(a, be, b_is_evals=True),
                    plan.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):
            assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), (q, n)


def _prime_below(limit, step):
    """largest prime q < limit with q = 1 (mod step)"""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2
            r += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    k = (limit - 2) // step
    while not is_prime(k * step + 1):
        k -= 1
    return k * step + 1


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [28, 29, 30])
def test_small_modulus_kernels_between_2_27_and_2_30(pkg, oracle, need_gpu, bits):
    """round 3: moduli between 2^32 / 25 and 2^30 run in 32-bit words too, on Harvey's butterflies (4 q fits the word; one
    conditional subtraction per butterfly instead of none).  The largest NTT-friendly prime below 2^bits — 4 q within
    a few thousand of 2^32 at bits = 30 — at every kernel shape: 256 threads x W polynomials, the one-workgroup sizes
    8192 / 16384, the two-pass sizes; forward, inverse, product with every evals combination, extreme inputs."""
    q = _prime_below(1 << bits, 1 << 18)
    assert 25 * q >= 1 << 32 and 4 * q < 1 << 32
    for n in (256, 2048, 4096, 8192, 16384, 1 << 15, 1 << 16, 1 << 17):
        a = _extreme_rows(oracle, q, n, 5000 + n)
        b = a[::-1].copy()
        plan = pkg.Plan(q, n)
        assert plan.arithmetic() == (3 if os.environ.get("FHE_EXT32", "1")[:1] != "0" else 1)
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n)
        want = oracle.rq_mul(q, n, a, b)
        ae, be = want[2], want[3]
        for got in (plan.rq_mul(a, b), plan.rq_mul(ae, b, a_is_evals=True), plan.rq_mul(a, be, b_is_evals=True),
                    plan.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):
            assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), (q, n)
    # 2^30 itself is past the form: 61-bit kernels, same words
    q31 = _prime_below(1 << 31, 1 << 18)
    assert pkg.Plan(q31, 4096).arithmetic() == 1
    a = _extreme_rows(oracle, q31, 4096, 6000)
    assert np.array_equal(pkg.Plan(q31, 4096).forward(a).reshape(-1), oracle.ntt(q31, 4096, a).reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [1 << 27, 1 << 30])
def test_small_modulus_n_2_18(pkg, oracle, need_gpu, limit):
    """round 3: n = 2^18 at a small modulus — six element-wise strided stages (64 registers per thread) + 4096-point blocks
    — in both butterfly forms; forward, inverse, product with cached evals"""
    q, n = _prime_below(limit, 1 << 19), 1 << 18
    a = _extreme_rows(oracle, q, n, 7000)[1:5]
    b = a[::-1].copy()
    plan = pkg.Plan(q, n)
    assert plan.arithmetic() == (3 if os.environ.get("FHE_EXT32", "1")[:1] != "0" else 1)
    A = plan.forward(a)
    assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), q
    assert np.array_equal(plan.inverse(A).reshape(-1), a.reshape(-1)), q
    want = oracle.rq_mul(q, n, a, b)
    for got in (plan.rq_mul(a, b), plan.rq_mul(want[2], b, a_is_evals=True)):
        assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), q


@pytest.mark.gpu
def test_release_stream_workspace_frees_every_slot(pkg, oracle, need_gpu):
    """the small-modulus two-pass sizes keep their u32 intermediates in a library workspace of their own slot;
    fhe_ntt_release_stream_workspace must hand that one back too (it used to free slots 0 and 1 only)"""
    import torch

    B, L = pkg.binding, pkg.load_library()
    q, n, batch = 786433, 65536, 6
    plan = pkg.Plan(q, n)
    a = oracle.fill_synthetic(q, 11, 0, batch * n)
    want = oracle.ntt(q, n, a)
    s1 = torch.cuda.Stream()
    held0 = L.fhe_ntt_workspace_bytes()
    x = torch.from_numpy(a.view(np.int64).copy()).cuda()
    y = torch.empty_like(x)
    c = torch.empty_like(x)
    torch.cuda.synchronize()
    plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, s1.cuda_stream)
    B._check(L.fhe_rq_mul_dev(plan.handle, x.data_ptr(), 0, x.data_ptr(), 0, c.data_ptr(), None, None, None, batch, None, s1.cuda_stream))
    s1.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint64), want)
    held1 = L.fhe_ntt_workspace_bytes()
    if plan.arithmetic() == 3:
        assert held1 >= held0 + batch * n * 4            # at least the u32 intermediate of the transform
    assert held1 > held0
    B._check(L.fhe_ntt_release_stream_workspace(s1.cuda_stream))
    assert L.fhe_ntt_workspace_bytes() == held0          # everything this stream took, whatever the slot
    plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, s1.cuda_stream)      # and the stream is still usable
    s1.synchronize()
    assert np.array_equal(y.cpu().numpy().view(np.uint64), want)
    B._check(L.fhe_ntt_release_stream_workspace(s1.cuda_stream))


def test_version_string_and_ablation_guard(pkg):
    """a production build never reports ABLATED; the binding refuses a library that does (tools/abl_build.sh) unless
    FHE_NTT_ALLOW_ABLATED=1 — checked on the loader's own logic, no second library needed"""
    v = pkg.load_library().fhe_ntt_version()
    assert v.startswith(b"fhe_ntt") and b"ABLATED" not in v
    import inspect

    src = inspect.getsource(pkg.binding.load_library)
    assert "ABLATED" in src and "FHE_NTT_ALLOW_ABLATED" in src


# ---- BFV: the tensor's scale-and-round in integers --------------------------------------------------


def _scale_round_int(N, q):
    """bfv32.hip: zq_scale_round_int, in Python integers"""
    a, f = divmod(N, q)
    return (a + (1 if 2 * f >= q else 0)) % q


def test_integer_scale_round_equals_the_f64_form_where_the_kernel_uses_it():
    """Zq::from_f64(round(num * v / q)) (ring_n.rs:130-138, zq.rs:32-39) in IEEE f64 — what the reference computes —
    against the integer form the tensor kernel uses when num * v < 2^52: every residue class around the half-integer
    boundaries (2f = q - 1, q, q + 1), the largest admitted N, and random values"""
    rnd = random.Random(0xF4E5_0303)
    for q in (65537, 12289, 786433, 1021, 2, 4, 6, 1 << 20, (1 << 21) - 9):
        cases = []
        for k in (0, 1, 2, 1000, (1 << 52) // q - 2, rnd.randrange((1 << 52) // q - 2)):
            for f in {0, 1, (q - 1) // 2 - 1, (q - 1) // 2, q // 2, q // 2 + 1, (q + 1) // 2, q - 1}:
                if 0 <= f < q:
                    cases.append(k * q + f)
        cases += [(1 << 52) - 1, (1 << 52) - q, (1 << 50) - 1, (1 << 50) + 1] + [rnd.randrange(1 << 52) for _ in range(1500)]
        N = np.array([c for c in cases if c < (1 << 52)], dtype=np.uint64)
        x = np.round(np.float64(1.0) * N.astype(np.float64) / np.float64(q))           # N is exact in f64: N < 2^53
        # np.round is half-to-even: the reference's f64::round is half away from zero — they differ only ON a half-integer
        frac_half = (2 * (N % np.uint64(q)) == np.uint64(q))
        x = np.where(frac_half, np.floor(N.astype(np.float64) / np.float64(q)) + 1.0, x)
        want = (x.astype(np.int64) % q).astype(np.uint64)
        got = np.array([_scale_round_int(int(v), q) for v in N], dtype=np.uint64)
        assert np.array_equal(got, want), q


def test_zq_from_f64_in_f64_alone_is_the_same_residue():
    """bfv32.hip: zq_from_f64_small — e = round(ef) exact, k = rint(e * fl(1/q)) within one of the quotient, r = e - k q
    adjusted into [0, q) — against ((e % q) + q) % q (zq.rs:32-39) for |e| < 2^50, negative values included"""
    rnd = random.Random(0xF4E5_0304)
    for q in (65537, 12289, 786433, 1021, 3, (1 << 21) - 9, (1 << 31) - 1):
        es = [0, 1, -1, q, -q, q - 1, 1 - q, (1 << 50) - 1, -(1 << 50) + 1] + [rnd.randrange(-(1 << 50), 1 << 50) for _ in range(3000)] + \
             [k * q + d for k in (1, 7, (1 << 50) // q - 1) for d in (-1, 0, 1)]
        e = np.array(es, dtype=np.float64)
        assert all(float(x) == y for x, y in zip(es, e))                      # exact integers in f64
        k = np.rint(e * np.float64(1.0 / q))
        for ei, ki in zip(es, k):
            r = ei - int(ki) * q                                               # the fma is exact: |k q| < 2^52
            assert -3 * q < 2 * r < 3 * q
            r = r + q if r < 0 else r
            r = r + q if r < 0 else r
            r = r - q if r >= q else r
            assert r == ei % q, (q, ei)
        if q < 1 << 30:
            # round 4: k = floor(e * fl(1/q)), r = e - k q in (-q, 2q), fixed up as a 32-bit integer: r + (q if r < 0), then the
            # unsigned min(r, r - q) — the tail the kernels run for q < 2^30
            kf = np.floor(e * np.float64(1.0 / q))
            for ei, ki in zip(es, kf):
                r = ei - int(ki) * q
                assert -q < r < 2 * q and -(1 << 31) <= r < 1 << 31
                r32 = (r + (q if r < 0 else 0)) & 0xffffffff
                assert min(r32, (r32 - q) & 0xffffffff) == ei % q, (q, ei)


@pytest.mark.gpu
def test_bfv_epilogue_forms_give_the_same_words(pkg, oracle, need_gpu):
    """The tensor / relinearisation epilogues have three forms of Zq::from_f64(round(num * v / den)): f64 with the
    general saturating conversion (FHE_BFV_SMALL_F64=0; the default until round 4), f64 alone (the default since, where
    the scaled coefficients stay below 2^50), and the integer form (FHE_BFV_INT_ROUND=1, tensor only).  Identical ciphertext words on the reference's
    parameters and on config 3's, where ~2/q of the coefficients sit next to a half-integer boundary; RLWE::mul
    covers the relinearisation's epilogue."""
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "import fhe_study_amd as pkg\n"
        "B = pkg.binding\n"
        "for q, n, t, batch in ((65537, 8192, 2, 24), (65537, 1024, 2, 9), (12289, 2048, 3, 5), (786433, 4096, 2, 4)):\n"
        "    rng = np.random.default_rng(q + n)\n"
        "    ab = [rng.integers(0, q, (batch, n), dtype=np.uint64) for _ in range(4)]\n"
        "    ab[0][0, :] = q - 1; ab[2][0, :] = q - 1; ab[1][0, :] = q - 1; ab[3][0, :] = q - 1\n"
        "    c = B.bfv_tensor(q, n, t, *ab)\n"
        "    pq = q * q * q\n"
        "    rlk = rng.integers(0, pq, (2, n), dtype=np.uint64)\n"
        "    o = B.bfv_mul(q, n, t, pq, rlk[0], rlk[1], *ab)\n"
        "    h = hashlib.sha256()\n"
        "    for x in list(c) + list(o): h.update(np.ascontiguousarray(x).tobytes())\n"
        "    print('digest', q, n, h.hexdigest())\n" % ROOT)
    outs = {}
    # round 4: the quotient as reciprocal + two fma, canonical source words taken as their own residues — and both switched off
    for name, extra in (("general", {"FHE_BFV_SMALL_F64": "0"}), ("f64", {"FHE_BFV_SMALL_F64": "1"}), ("int", {"FHE_BFV_INT_ROUND": "1"}),
                        ("ieee-division", {"FHE_BFV_FAST_DIV": "0", "FHE_BFV_BELOW_P": "0"}),
                        ("ieee-division-general", {"FHE_BFV_FAST_DIV": "0", "FHE_BFV_SMALL_F64": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, name + r.stdout + r.stderr
        outs[name] = [l for l in r.stdout.splitlines() if l.startswith("digest")]
        assert len(outs[name]) == 4
    assert outs["general"] == outs["f64"] == outs["int"] == outs["ieee-division"] == outs["ieee-division-general"]


# ---- the top of the reference's modulus range: 2^62 <= q < 2^63 (AR = 3 of ntt_kernels.hip; csrc/generic63.hip) ----------
# Zq::add computes `self.v + rhs.v` in a u64 (arith/src/zq.rs:225): the reference works for every q below 2^63.  The lazy
# kernels need 4q < 2^64; this range runs strict butterflies (round 5: in the two-pass / fused kernels; before: plain launches).
Q63_TOP = 9223372036844421121      # largest prime = 1 (mod 2^17) below 2^63
Q63_BOTTOM = 4611686018429485057   # smallest one above 2^62


def test_plans_exist_up_to_the_reference_limit(pkg, oracle):
    B = pkg.binding
    for q in (Q63_TOP, Q63_BOTTOM):
        assert q >> 62 == 1 and (q - 1) % (1 << 17) == 0
        for n in (2, 16, 1024):
            plan = pkg.Plan(q, n)
            r, ri = plan.tables()
            orr, ori, n_inv, psi = oracle.roots(q, n)
            assert np.array_equal(r, orr) and np.array_equal(ri, ori)
            assert plan.info() == dict(q=q, n=n, psi=psi, n_inv=n_inv)
            assert plan.arithmetic() == 4 and pkg.Plan.ARITH_NAMES[4] == "strict63"
            assert pkg.load_library().fhe_glwe_ksk_prepared_words(plan.handle, 1, 2, 8) == 1 * 8 * 2 * n    # forward transforms modulo q
    with pytest.raises(pkg.FheError) as ei:
        pkg.Plan((1 << 63) + 2 * 1024 + 1, 1024)       # where the reference's Zq::add would overflow
    assert ei.value.code == B.FHE_E_BAD_Q


def test_strict_arithmetic_never_leaves_a_word():
    """the bounds generic63.hip and reduce128_63 (zq_device.hpp) rely on, at the extremes of the range"""
    M = 1 << 64
    for q in (Q63_TOP, Q63_BOTTOM, (1 << 63) - 1, 1 << 62):
        onep, r64 = M // q, M % q
        r64p = (r64 << 64) // q
        for y in (0, 1, q - 1, q, 2 * q - 1, M - 1, M - q, (1 << 63)):
            for w in (0, 1, q - 1, q // 2, r64):
                wp = (w << 64) // q
                r = y * w - ((y * wp) >> 64) * q            # mul_shoup_lazy: exact, and below 2q < 2^64
                assert 0 <= r < 2 * q < M and r % q == y * w % q
            lo = y - ((y * onep) >> 64) * q                 # reduce_any / the low half of reduce128_63
            assert 0 <= lo < 2 * q and lo % q == y % q
        # canon2's sign test: x < 2q must satisfy x < 2^63 + q; and a canonical sum / difference stays below 2q
        assert 2 * q <= (1 << 63) + q
        for x, y in ((q - 1, q - 1), (0, q - 1), (q - 1, 0), (0, 0)):
            assert x + y < 2 * q and 0 < x + (q - y) <= 2 * q - 1 + (y == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [Q63_TOP, Q63_BOTTOM])
def test_moduli_between_2_62_and_2_63(pkg, oracle, need_gpu, q):
    """forward, inverse, the product with every evals combination and the pointwise product against the oracle, at every
    grouping of the stages (log2 n = 1 .. 16: short first group of 1, 2, 3 stages or none), on inputs that sit at the
    top of the range"""
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096, 8192, 1 << 14, 1 << 16):
        a = _extreme_rows(oracle, q, n, 6300 + n)
        b = a[::-1].copy()
        plan = pkg.Plan(q, n)
        assert plan.arithmetic() == 4
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n)
        assert np.array_equal(plan.inverse(A).reshape(-1), a.reshape(-1)), (q, n)
        want = oracle.rq_mul(q, n, a, b)
        ae, be = want[2], want[3]
        for got in (plan.rq_mul(a, b), plan.rq_mul(ae, b, a_is_evals=True), plan.rq_mul(a, be, b_is_evals=True),
                    plan.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):
            assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), (q, n)
        pw = plan.pointwise_mul(a, b).reshape(-1)
        assert np.array_equal(pw, oracle.pointwise_mul(q, a.reshape(-1), b.reshape(-1))), (q, n)
        if n <= 256:
            assert pw.tolist() == [int(x) * int(y) % q for x, y in zip(a.reshape(-1), b.reshape(-1))]


@pytest.mark.gpu
def test_elementwise_rows_at_2_63(pkg, oracle, need_gpu):
    """Zq add / sub / neg / scalar product (zq.rs:219-328) in Python integers (the keyed rows at this modulus:
    tests/test_glue_rows.py, the Q63 cases)"""
    import torch

    q, n, batch = Q63_TOP, 64, 3
    L, B, chk = pkg.load_library(), pkg.binding, pkg.binding._check
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(63)
    a = rng.integers(0, q, (batch, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, n), dtype=np.uint64)
    a[0, :5] = [0, q - 1, 1, q // 2, q - 1]
    b[0, :5] = [0, q - 1, q - 1, q // 2 + 1, 1]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
    da, db = dev(a), dev(b)
    dc = torch.empty_like(da)
    got = lambda: dc.cpu().numpy().view(np.uint64).reshape(-1).tolist()
    A, Bv = [int(x) for x in a.reshape(-1)], [int(x) for x in b.reshape(-1)]
    chk(L.fhe_rq_add_dev(plan.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, None))
    assert got() == [(x + y) % q for x, y in zip(A, Bv)]
    chk(L.fhe_rq_sub_dev(plan.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, None))
    assert got() == [(x - y) % q for x, y in zip(A, Bv)]
    chk(L.fhe_rq_neg_dev(plan.handle, da.data_ptr(), dc.data_ptr(), batch, None))
    assert got() == [(-x) % q for x in A]
    for s in (0, 1, q - 1, q, (1 << 64) - 1, 0x9E3779B97F4A7C15):
        chk(L.fhe_rq_mul_by_u64_dev(plan.handle, da.data_ptr(), s, dc.data_ptr(), batch, None))
        assert got() == [x * (s % q) % q for x in A], s


@pytest.mark.gpu
def test_device_entry_points_at_2_63_in_place_ragged_and_with_evals(pkg, oracle, need_gpu):
    """the device entry points at the top of the range: transforms in place, a batch that ends inside a workgroup,
    the product with the library workspace writing c over a and keeping C's evals, and the synthetic fill"""
    import torch

    q, st = Q63_TOP, torch.cuda.current_stream().cuda_stream
    for n, batch in ((64, 37), (8192, 5), (1 << 15, 3)):
        plan = pkg.Plan(q, n)
        x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
        pkg.binding.fill_synthetic_dev(q, 0x63, 0, batch * n, x.data_ptr(), st)
        a = x.cpu().numpy().view(np.uint64)
        assert np.array_equal(a, oracle.fill_synthetic(q, 0x63, 0, batch * n)) and int(a.max()) < q
        y = x.clone()
        plan.forward_dev(y.data_ptr(), y.data_ptr(), batch, st)
        assert np.array_equal(y.cpu().numpy().view(np.uint64), oracle.ntt(q, n, a).reshape(-1)), n
        plan.inverse_dev(y.data_ptr(), y.data_ptr(), batch, st)
        assert torch.equal(y, x), n
        b = torch.empty_like(x)
        pkg.binding.fill_synthetic_dev(q, 0x64, 0, batch * n, b.data_ptr(), st)
        ce = torch.empty_like(x)
        c = x.clone()
        plan.rq_mul_dev(c.data_ptr(), b.data_ptr(), c.data_ptr(), batch, d_c_evals=ce.data_ptr(), stream=st)   # c over a
        oc, oce, _, _ = oracle.rq_mul(q, n, a, b.cpu().numpy().view(np.uint64))
        assert np.array_equal(c.cpu().numpy().view(np.uint64), oc.reshape(-1)), n
        assert np.array_equal(ce.cpu().numpy().view(np.uint64), oce.reshape(-1)), n


# ---- differential sweep over the modulus range ----------------------------------------------------------------------------
def _random_cases():
    """(q, n) pairs drawn once (fixed seed): a prime = 1 (mod 2n) near a random point of every bit length from 17 to 63, plus
    primes hugging the boundaries where the engine changes arithmetic (2^32/25, 2^30, 2^32, 2^61 and the pseudo-Mersenne
    eligibility limit delta <= 2^(k-39), 2^62, 2^63)"""
    rng = random.Random(20260403)
    cases = []
    for bits in range(17, 64):
        n = 1 << rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14])
        lo, hi = 1 << (bits - 1), 1 << bits
        cases.append((_prime_below(rng.randrange(lo + (hi - lo) // 8, hi), 2 * n), n))
    for limit in ((1 << 32) // 25, (1 << 32) // 25 + (1 << 22), 1 << 30, (1 << 30) + (1 << 24), 1 << 32, (1 << 32) + (1 << 26),
                  1 << 61, (1 << 61) + (1 << 50), 1 << 62, (1 << 62) + (1 << 40), 1 << 63):
        for n in (256, 4096):
            cases.append((_prime_below(limit, 2 * n), n))
    for k in (56, 58, 60, 61):                      # either side of the pseudo-Mersenne eligibility limit
        for mult in (1, 3):
            cases.append((_prime_below((1 << k) - mult * (1 << (k - 40)), 1 << 13), 4096))
    return sorted(set(cases))


def test_differential_cases_cover_every_arithmetic(pkg):
    """host-side: the sweep below reaches all five arithmetic forms (fhe_ntt_plan_arithmetic)"""
    forms = {pkg.Plan(q, n).arithmetic() for q, n in _random_cases()}
    ext_on = os.environ.get("FHE_EXT32", "1")[:1] != "0"
    pm_on = os.environ.get("FHE_PM", "1")[:1] != "0"
    assert forms >= ({0, 1, 4} | ({3} if ext_on else set()) | ({2} if pm_on else set())), forms


@pytest.mark.gpu
def test_differential_sweep_over_the_modulus_range(pkg, oracle, need_gpu):
    """forward, inverse, product (with its three evals) and pointwise product against the oracle for every case of
    _random_cases(): three rows each — all q-1, alternating, pseudo-random"""
    for q, n in _random_cases():
        a = _extreme_rows(oracle, q, n, q % 1009)[[1, 2, 4]]
        b = _extreme_rows(oracle, q, n, q % 1013)[[4, 1, 5]]
        plan = pkg.Plan(q, n)
        A = plan.forward(a)
        assert np.array_equal(A.reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n, plan.arithmetic())
        assert np.array_equal(plan.inverse(a).reshape(-1), oracle.intt(q, n, a).reshape(-1)), (q, n, plan.arithmetic())
        got, want = plan.rq_mul(a, b), oracle.rq_mul(q, n, a, b)
        assert all(np.array_equal(x.reshape(-1), y.reshape(-1)) for x, y in zip(got, want)), (q, n, plan.arithmetic())
        assert np.array_equal(plan.pointwise_mul(a, b).reshape(-1), oracle.pointwise_mul(q, a.reshape(-1), b.reshape(-1))), (q, n)
