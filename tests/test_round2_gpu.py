"""GPU tests added in round 2: the full-size BASELINE.json configurations, the API holes closed
(per-stream workspaces, opt-in canonical check, plan preparation) and the multi-rank driver run
as real processes.  Everything is compared word for word with the CPU oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import Q16, Q61, ROOT

pytestmark = pytest.mark.gpu
U64 = 1 << 64


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(pkg):
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


# ---- BASELINE.json configs at FULL size ----------------------------------------------------------

def test_config5_full_size_n65536_batch65536(pkg, oracle):
    """configs[4] on one GPU exactly as bench.py runs it: N = 2^16, q61, 65536 polynomials (32 GiB in,
    32 GiB out).  Round trip bit-exact over all 2^32 words; forward == oracle on the first 8, last 8 and
    48 pseudo-random rows; outputs canonical."""
    import torch

    q, n, batch = Q61, 65536, 65536
    plan = pkg.Plan(q, n)
    st = torch.cuda.current_stream().cuda_stream
    seed = 0xF4E50005
    x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    y = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    pkg.binding.fill_synthetic_dev(q, seed, 0, batch * n, x.data_ptr(), st)
    plan.forward_dev(x.data_ptr(), y.data_ptr(), batch, st)
    rows = sorted(set(list(range(8)) + list(range(batch - 8, batch)) +
                      [int(r) for r in np.random.default_rng(7).integers(0, batch, 48)]))
    Y = y.view(batch, n)
    for r in rows:
        a = oracle.fill_synthetic(q, seed, r * n, n)
        assert np.array_equal(_u64(Y[r]), oracle.ntt(q, n, a)), f"row {r}"
    # canonical: 0 <= y < q as unsigned words (q < 2^63, so the int64 view is order-preserving)
    chunk = 1 << 28
    for i in range(0, batch * n, chunk):
        part = y[i:i + chunk]
        assert int(part.min()) >= 0 and int(part.max()) < q
    plan.inverse_dev(y.data_ptr(), y.data_ptr(), batch, st)        # in place, back to coefficients
    for i in range(0, batch * n, chunk):
        assert torch.equal(y[i:i + chunk], x[i:i + chunk]), f"round trip differs in words [{i}, {i + chunk})"


def test_config4_630_external_products_sharded_like_8_gpus(pkg, oracle):
    """configs[3]: 630 TGGSW x TGLWE external products (N=1024, k=1, l=64), block-sharded 79 x 7 + 77 by
    fhe_shard_range as 8 ranks would; here the 8 shards run one after another on one GPU.  The
    concatenation equals the unsharded run, and 8 products (first / last of a shard, shard boundaries)
    equal the oracle's schoolbook."""
    import torch

    n, k, l, total, world = 1024, 1, 64, 630, 8
    L, B = pkg.load_library(), pkg.binding
    rng = np.random.default_rng(44)
    g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
    c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (total, k + 1, n), dtype=np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    whole = torch.empty_like(c)
    B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), whole.data_ptr(), total, st))
    parts, sizes = [], []
    for r in range(world):
        b0, b1 = B.shard_range(total, world, r)
        sizes.append(b1 - b0)
        out = torch.empty((b1 - b0, k + 1, n), dtype=torch.int64, device="cuda")
        shard = c[b0:b1].contiguous()
        B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), shard.data_ptr(), out.data_ptr(), b1 - b0, st))
        parts.append(out)
    assert sizes == [79] * 7 + [77]
    got = torch.cat(parts)
    assert torch.equal(got, whole)
    check = [0, 78, 79, 315, 552, 553, 628, 629]
    want = oracle.external_product(n, k, l, _u64(g), _u64(c[check]))
    assert np.array_equal(_u64(got[check]), want)


def test_config3_bfv_multiply_n8192_batch(pkg, oracle):
    """configs[2] at the batch bench.py times (256 ciphertext pairs, N = 8192): every pair of the batch is
    the same function of its inputs (rows permuted in -> rows permuted out), and two pairs equal the
    oracle's schoolbook tensor + relinearisation word for word."""
    import torch

    q, n, t, batch = Q16, 8192, 2, 256
    pq = q * q * q
    L, B = pkg.load_library(), pkg.binding
    rng = np.random.default_rng(33)
    ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).cuda()
    rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).cuda()
    out = torch.empty((2, batch, n), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), out.data_ptr(), batch, st))
    perm = torch.from_numpy(np.random.default_rng(5).permutation(batch)).cuda()
    ab2 = ab[:, perm].contiguous()
    out2 = torch.empty_like(out)
    B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab2.data_ptr(), out2.data_ptr(), batch, st))
    assert torch.equal(out2, out[:, perm])
    a, r = _u64(ab), _u64(rlk)
    for i in (0, batch - 1):
        w0, w1 = oracle.bfv_mul(q, n, t, pq, r[0], r[1], a[0, i:i + 1], a[1, i:i + 1], a[2, i:i + 1], a[3, i:i + 1])
        assert np.array_equal(_u64(out[0, i]), w0[0]) and np.array_equal(_u64(out[1, i]), w1[0])


# ---- the reference's round-trip loop with fresh randomness ------------------------------------------

def test_intt_ntt_identity_1000_fresh_polynomials_n512(pkg):
    """arith/src/ntt.rs:217-234 draws 1000 polynomials from an UNSEEDED generator; restated with
    fresh entropy on every run (the seed is printed on failure), through the HIP path."""
    seed = int.from_bytes(os.urandom(8), "little")
    rng = np.random.default_rng(seed)
    plan = pkg.Plan(Q16, 512)
    for _ in range(4):                                   # 4 x 250 = 1000, as separate calls
        a = rng.integers(0, Q16, size=(250, 512), dtype=np.uint64)
        assert np.array_equal(plan.inverse(plan.forward(a)).reshape(a.shape), a), f"seed {seed}"


# ---- API holes -----------------------------------------------------------------------------------------

def test_two_streams_share_no_workspace(pkg, oracle):
    """*_dev entry points that use the library workspace, issued on two streams with no
    synchronisation between them: each stream has its own workspace, so both results are exact.
    (Round 1 kept one buffer per device: this test then fails with silently wrong words.)"""
    import torch

    L, B = pkg.load_library(), pkg.binding
    q, n, batch = Q61, 16384, 24                          # two-pass size: fhe_rq_mul_dev needs scratch
    plan = pkg.Plan(q, n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ops = []
    for i, s in enumerate((s1, s2)):
        a = oracle.fill_synthetic(q, 100 + i, 0, batch * n)
        b = oracle.fill_synthetic(q, 200 + i, 0, batch * n)
        da = torch.from_numpy(a.view(np.int64).copy()).cuda()
        db = torch.from_numpy(b.view(np.int64).copy()).cuda()
        ops.append((s, da, db, torch.empty_like(da), oracle.rq_mul(q, n, a, b)[0]))
    # external products on the same two streams, interleaved with the ring products
    nn, k, l, eb = 1024, 1, 64, 16
    rng = np.random.default_rng(9)
    g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, nn), dtype=np.int64)).cuda()
    ext = []
    for i, s in enumerate((s1, s2)):
        c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (eb, k + 1, nn), dtype=np.int64)).cuda()
        ext.append((s, c, torch.empty_like(c), oracle.external_product(nn, k, l, _u64(g), _u64(c))))
    torch.cuda.synchronize()
    for rep in range(6):
        for (s, da, db, dc, _), (_, c, eo, _) in zip(ops, ext):
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, stream=s.cuda_stream)
            B._check(L.fhe_tggsw_external_product_dev(nn, k, l, g.data_ptr(), c.data_ptr(), eo.data_ptr(), eb, s.cuda_stream))
    torch.cuda.synchronize()
    for (_, _, _, dc, want), (_, _, eo, ewant) in zip(ops, ext):
        assert np.array_equal(_u64(dc), want)
        assert np.array_equal(_u64(eo), ewant)
    # a stream about to be destroyed returns its workspaces; the library keeps working on it and on the other one
    B._check(L.fhe_ntt_release_stream_workspace(s1.cuda_stream))
    for (s, da, db, dc, want) in ops:
        dc.zero_()
        plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, stream=s.cuda_stream)
    torch.cuda.synchronize()
    for (_, _, _, dc, want) in ops:
        assert np.array_equal(_u64(dc), want)


def test_opt_in_canonical_check_rejects_what_the_reference_cannot_construct(pkg, oracle):
    """FHE_NTT_CHECK_CANONICAL (here through fhe_ntt_set_check_canonical): a value >= q handed to a
    transform comes back as FHE_E_NOT_CANONICAL; canonical inputs are unaffected; switched off
    (the default) the call returns FHE_OK as before."""
    B = pkg.binding
    q, n = Q61, 4096
    plan = pkg.Plan(q, n)
    good = oracle.fill_synthetic(q, 1, 0, 3 * n)
    bad = good.copy()
    bad[2 * n + 17] = q                                    # the smallest non-canonical value
    B.set_check_canonical(True)
    try:
        assert np.array_equal(plan.forward(good), oracle.ntt(q, n, good))
        for call in (lambda: plan.forward(bad), lambda: plan.inverse(bad), lambda: plan.rq_mul(good, bad),
                     lambda: plan.rq_mul(bad, good), lambda: plan.pointwise_mul(bad, good)):
            with pytest.raises(pkg.FheError) as ei:
                call()
            assert ei.value.code == B.FHE_E_NOT_CANONICAL
    finally:
        B.set_check_canonical(False)
    plan.forward(bad)                                      # undefined words, but FHE_OK: the documented default
    assert np.array_equal(plan.forward(good), oracle.ntt(q, n, good))


def test_plan_prepare_makes_the_first_transform_capturable(pkg, oracle):
    """fhe_ntt_plan_prepare uploads the tables ahead of time, so the very first transform of a plan
    can already sit inside a stream capture (no allocation, copy or synchronisation in it)."""
    import torch

    L = pkg.load_library()
    q, n, batch = 12289, 2048, 4                            # a plan no other test creates
    plan = pkg.Plan(q, n)
    pkg.binding._check(L.fhe_ntt_plan_prepare(plan.handle))
    a = oracle.fill_synthetic(q, 77, 0, batch * n)
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    out = torch.empty_like(da)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.forward_dev(da.data_ptr(), out.data_ptr(), batch, torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(_u64(out), oracle.ntt(q, n, a))


# ---- the multi-rank driver, run as real processes ---------------------------------------------------------

def test_bench_two_ranks_on_one_gpu(pkg, oracle, tmp_path):
    """bench.py under torch.distributed.run with two ranks sharing cuda:0 (gloo for the
    rendezvous/barriers, as a 1-GPU box allows): exercises process-group setup, the barrier + MAX
    timing, rank-offset input generation and the parity leg of BOTH ranks; n_gpus must read 2."""
    from test_sharding_gloo import _free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--share-gpu", "--backend", "gloo", "--batch-per-gpu", "256", "--no-cpu-baseline",
           "--parity-all-ranks", "--gather-check"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512
    assert out["parity"]["mismatching_rows"] == 0
    assert out["parity"]["ranks_checked"] == 2
    assert out["gather_check"]["equal_to_single_rank_transform"] is True


# ---- the product at two-pass sizes: fused middle kernel ------------------------------

@pytest.mark.parametrize("q,log_n,batch", [(Q61, 14, 19), (Q61, 15, 5), (Q61, 16, 17), (Q61, 17, 3), (Q61, 18, 2),
                                           (Q61, 20, 1), (Q16, 14, 18), (4611686018425815041, 14, 3)])
def test_two_pass_product_every_evals_combination(pkg, oracle, q, log_n, batch):
    """n >= 2^14: strided(a) | strided(b) in one launch, ONE middle kernel (contiguous forward of both
    operands, pointwise product, contiguous inverse), strided inverse.  Every combination of cached
    evals (ring_nq.rs:590-599) and every optional output, ragged batches (W = 16 .. 1 units per
    workgroup), a tiny and a 62-bit modulus; word for word against the oracle."""
    n = 1 << log_n
    a = oracle.fill_synthetic(q, 5 + log_n, 0, batch * n)
    b = oracle.fill_synthetic(q, 6 + log_n, 0, batch * n)
    want = oracle.rq_mul(q, n, a, b)                     # c, c_evals, a_evals, b_evals
    c, ce, ae, be = want
    P = pkg.Plan(q, n)
    for got in (P.rq_mul(a, b), P.rq_mul(ae, b, a_is_evals=True), P.rq_mul(a, be, b_is_evals=True),
                P.rq_mul(ae, be, a_is_evals=True, b_is_evals=True)):
        for x, y in zip(got, want):
            assert np.array_equal(np.asarray(x).reshape(-1), y.reshape(-1)), (q, n)
    assert np.array_equal(P.rq_mul(a, b, want_evals=False)[0].reshape(-1), c.reshape(-1))


def test_two_pass_product_device_entry_point_outputs_workspaces_and_tiles(pkg, oracle):
    """fhe_rq_mul_dev at n = 2^16: library workspace and caller workspace, each optional output on its
    own, the product written over an operand (c == a), and the batch walked in tiles."""
    import torch

    q, n, batch = Q61, 65536, 21
    plan = pkg.Plan(q, n)
    st = torch.cuda.current_stream().cuda_stream
    a = oracle.fill_synthetic(q, 71, 0, batch * n)
    b = oracle.fill_synthetic(q, 72, 0, batch * n)
    c, ce, ae, be = oracle.rq_mul(q, n, a, b)
    dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    da, db = dev(a), dev(b)
    work = torch.empty(plan.workspace_bytes(batch) // 8, dtype=torch.int64, device="cuda")
    for tile in (0, 4, 16):
        pkg.binding.set_batch_tile(tile)
        for w in (None, work.data_ptr()):
            dc, dce, dae, dbe = (torch.empty_like(da) for _ in range(4))
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, d_work=w, stream=st)
            assert np.array_equal(_u64(dc), c), (tile, w)
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, d_c_evals=dce.data_ptr(), d_work=w, stream=st)
            assert np.array_equal(_u64(dc), c) and np.array_equal(_u64(dce), ce)
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, d_a_evals=dae.data_ptr(), d_work=w, stream=st)
            assert np.array_equal(_u64(dc), c) and np.array_equal(_u64(dae), ae)
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, d_b_evals=dbe.data_ptr(), d_work=w, stream=st)
            assert np.array_equal(_u64(dc), c) and np.array_equal(_u64(dbe), be)
    pkg.binding.set_batch_tile(0)
    # in place over an operand, and over an operand that is evals
    x = da.clone()
    plan.rq_mul_dev(x.data_ptr(), db.data_ptr(), x.data_ptr(), batch, stream=st)
    assert np.array_equal(_u64(x), c)
    x = dev(ae)
    plan.rq_mul_dev(x.data_ptr(), db.data_ptr(), x.data_ptr(), batch, a_is_evals=True, stream=st)
    assert np.array_equal(_u64(x), c)
    # mul_mut (ring_nq.rs:564-583): the operands' buffers end up holding their evals
    x, y = da.clone(), db.clone()
    dc = torch.empty_like(da)
    plan.rq_mul_dev(x.data_ptr(), y.data_ptr(), dc.data_ptr(), batch, d_a_evals=x.data_ptr(), d_b_evals=y.data_ptr(), stream=st)
    assert np.array_equal(_u64(dc), c) and np.array_equal(_u64(x), ae) and np.array_equal(_u64(y), be)


# ---- decompose -> forward NTT -> multiply-accumulate in one kernel ------------------------

@pytest.mark.parametrize("n,k,l,batch", [(1024, 1, 64, 5),      # BASELINE.json configs[3] shape: 4 units per workgroup, 4 parts
                                         (256, 1, 64, 3),       # 16 units per workgroup
                                         (512, 1, 7, 9),        # l not a multiple of the units: steps straddle source rows
                                         (2048, 1, 33, 2),
                                         (4096, 1, 64, 1)])     # 512-thread workgroups of digit32.hip (61-bit family: 64 accumulators, the unfused form)
def test_fused_external_product_and_prepared_key(pkg, oracle, n, k, l, batch):
    """TGGSW x TGLWE through digit_mac_kernel (tfhe/src/tggsw.rs:45-62): same words as the oracle's
    schoolbook; the key prepared once (fhe_tggsw_prepare_dev) gives the same words again, for several
    batches, without touching the original key."""
    import torch

    L, B = pkg.load_library(), pkg.binding
    rng = np.random.default_rng(n + l)
    g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
    c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()
    c[0, 0, :4] = torch.tensor([0, -1, 1, -(1 << 63)], dtype=torch.int64)        # all-zero / all-one digit columns
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty_like(c)
    B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), out.data_ptr(), batch, st))
    want = oracle.external_product(n, k, l, _u64(g), _u64(c))
    assert np.array_equal(_u64(out), want)
    words = L.fhe_tggsw_prepared_words(n, k, l)
    assert words == 2 * (k + 1) * l * (k + 1) * n
    prep = torch.empty(words, dtype=torch.int64, device="cuda")
    B._check(L.fhe_tggsw_prepare_dev(n, k, l, g.data_ptr(), prep.data_ptr(), st))
    g.zero_()                                                                     # the prepared key stands alone
    for nb in (batch, 1):
        out2 = torch.empty((nb, k + 1, n), dtype=torch.int64, device="cuda")
        B._check(L.fhe_tggsw_external_product_prepared_dev(n, k, l, prep.data_ptr(), c.data_ptr(), out2.data_ptr(), nb, st))
        assert np.array_equal(_u64(out2), want[:nb])
    assert L.fhe_tggsw_prepared_words(16384, 1, 64) == 0                          # two-prime shape: no prepared form
    assert L.fhe_tggsw_prepare_dev(16384, 1, 64, prep.data_ptr(), prep.data_ptr(), st) == B.FHE_E_INVALID


@pytest.mark.parametrize("q,n,k,l,batch", [(Q61, 4096, 1, 61, 3),     # the shape tools/bench_next.py times: 1 unit, 16 positions per thread
                                           (Q61, 1024, 1, 61, 9),
                                           (Q61, 1024, 2, 33, 2),     # k+1 = 3 output rows
                                           (Q16, 256, 1, 16, 20),     # q = 65537 < 2^l: the saturation branch of Zq::decompose everywhere
                                           (Q61, 2048, 2, 64, 1),     # l = 64: `1 << l` wraps like a --release build
                                           (Q61, 2048, 1, 64, 2),     # the same on the two-small-prime kernels (k = 1), 8 positions per thread
                                           (Q61, 512, 1, 40, 5),
                                           (Q16, 4096, 1, 16, 2),     # 512-thread workgroups with every digit saturated
                                           (Q61, 256, 1, 7, 33),      # more ciphertexts than a tail workgroup holds rows, odd digit count
                                           (9223372036844421121, 1024, 1, 63, 3)])   # just below 2^63 (round 3): plain kernels, strict accumulators
def test_fused_key_switch(pkg, oracle, q, n, k, l, batch):
    """GLWE::key_switch (gfhe/src/glwe.rs:126-137) with base-2 decomposition through digit_mac_kernel;
    key in coefficients and key resident in the NTT domain (FHE_A_IS_EVALS) give the oracle's words."""
    import torch

    L, B = pkg.load_library(), pkg.binding
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(n + l + k)
    glwe = rng.integers(0, q, (batch, k + 1, n), dtype=np.uint64)
    glwe[0, 0, :4] = [0, 1, (1 << min(l, 62)) % q, q - 1]                         # both decompose branches
    ksk = rng.integers(0, q, (k, l, k + 1, n), dtype=np.uint64)
    dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    dglwe, dksk = dev(glwe), dev(ksk)
    dout = torch.empty_like(dglwe)
    B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, dglwe.data_ptr(), dksk.data_ptr(), dout.data_ptr(), batch, 0, None))
    want = np.empty((batch, k + 1, n), dtype=np.uint64)
    for i in range(batch):
        oracle.glue("key_switch", q, n, k, 2, l, glwe[i], ksk, want[i])
    assert np.array_equal(_u64(dout), want)
    dKSK, dout2 = torch.empty_like(dksk), torch.empty_like(dout)
    plan.forward_dev(dksk.data_ptr(), dKSK.data_ptr(), k * l * (k + 1))
    B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, dglwe.data_ptr(), dKSK.data_ptr(), dout2.data_ptr(), batch,
                                       B.FHE_A_IS_EVALS, None))
    assert np.array_equal(_u64(dout2), want)
    # resident key in the engine's own (opaque) form, built once; the original key may then go away
    words = L.fhe_glwe_ksk_prepared_words(plan.handle, k, 2, l)
    assert words in (ksk.size, 2 * ksk.size)
    prep = torch.empty(words, dtype=torch.int64, device="cuda")
    B._check(L.fhe_glwe_ksk_prepare_dev(plan.handle, k, 2, l, dksk.data_ptr(), prep.data_ptr(), None))
    assert L.fhe_glwe_ksk_prepare_dev(plan.handle, k, 2, l, dksk.data_ptr(), dksk.data_ptr(), None) == B.FHE_E_INVALID
    dksk.zero_()
    for nb in (batch, 1):
        dout3 = torch.zeros((nb, k + 1, n), dtype=torch.int64, device="cuda")
        B._check(L.fhe_glwe_key_switch_prepared_dev(plan.handle, k, 2, l, dglwe.data_ptr(), prep.data_ptr(), dout3.data_ptr(), nb, None))
        assert np.array_equal(_u64(dout3), want[:nb])


@pytest.mark.parametrize("q,n,batch", [(Q16, 256, 37), (Q16, 1024, 9), (Q16, 4096, 5), (12289, 2048, 3), (Q16, 512, 1),
                                        (Q16, 8192, 3), (Q16, 16384, 2), (786433, 8192, 2)])       # one workgroup of n / 16 threads
def test_small_modulus_transforms_in_32_bit_words(pkg, oracle, q, n, batch):
    """q < 2^32 / 25, 2^8 <= n <= 2^14: NTT::ntt / NTT::intt / Rq x Rq (arith/src/ntt.rs:44-104, ring_nq.rs:586-607) run
    in one 32-bit word per coefficient (smallq.hip) — the same canonical words, in the same order, as the oracle's; the
    kernel timer names prove the path; cached-evals products (61-bit kernels) consume its transforms unchanged."""
    import torch

    B = pkg.binding
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(q + n)
    a = rng.integers(0, q, (batch, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, n), dtype=np.uint64)
    a[0, :3] = [0, q - 1, 1]
    dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    da, db = dev(a), dev(b)
    A, r, c = torch.empty_like(da), torch.empty_like(da), torch.empty_like(da)
    B.kernel_timing_reset(); B.kernel_timing_enable(True)
    plan.forward_dev(da.data_ptr(), A.data_ptr(), batch)
    plan.inverse_dev(A.data_ptr(), r.data_ptr(), batch)
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c.data_ptr(), batch)
    torch.cuda.synchronize()
    names = set(B.kernel_timing_read())
    B.kernel_timing_enable(False)
    lg = n.bit_length() - 1
    if os.environ.get("FHE_EXT32", "1")[:1] != "0":            # FHE_EXT32=0 keeps these moduli on the 61-bit kernels: same words
        assert {f"sq_forward_{lg}", f"sq_inverse_{lg}", f"sq_rq_mul_{lg}"} <= names, names
    assert np.array_equal(_u64(A), oracle.ntt(q, n, a).reshape(batch, n))
    assert np.array_equal(_u64(r), a)
    assert np.array_equal(_u64(c), oracle.rq_mul(q, n, a, b)[0].reshape(batch, n))
    # the transforms feed the 61-bit product kernels as cached evals (ring_nq.rs:590-599)
    Bv, c2 = torch.empty_like(db), torch.empty_like(da)
    plan.forward_dev(db.data_ptr(), Bv.data_ptr(), batch)
    plan.rq_mul_dev(A.data_ptr(), Bv.data_ptr(), c2.data_ptr(), batch, a_is_evals=True, b_is_evals=True)
    assert torch.equal(c2, c)


@pytest.mark.parametrize("q,n,batch", [(Q16, 256, 19), (Q16, 4096, 3), (12289, 1024, 5), (Q16, 8192, 2), (Q16, 16384, 2),
                                        (Q16, 32768, 2), (786433, 65536, 1)])                      # two-pass sizes: the middle kernel
def test_small_modulus_product_with_cached_evals(pkg, oracle, q, n, batch):
    """ring_nq.rs:586-607 at a small modulus (sq_rq_mul_kernel, sq_big_rq_mul_kernel, sq2_block_mul_kernel): every combination of operands given as cached
    evals, with the three evals outputs — the words of the oracle's mul / mul_mut, through the 32-bit kernel (timer name)."""
    import torch

    B = pkg.binding
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(3 * q + n)
    a = rng.integers(0, q, (batch, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, n), dtype=np.uint64)
    wc, wce, wae, wbe = (x.reshape(batch, n) for x in oracle.rq_mul(q, n, a, b))
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64).copy()).cuda()
    for a_ev in (False, True):
        for b_ev in (False, True):
            da, db = dev(wae if a_ev else a), dev(wbe if b_ev else b)
            c, ce, ae, be = (torch.zeros((batch, n), dtype=torch.int64, device="cuda") for _ in range(4))
            B.kernel_timing_reset(); B.kernel_timing_enable(True)
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c.data_ptr(), batch, a_is_evals=a_ev, b_is_evals=b_ev,
                            d_c_evals=ce.data_ptr(), d_a_evals=ae.data_ptr(), d_b_evals=be.data_ptr())
            torch.cuda.synchronize()
            assert os.environ.get("FHE_EXT32", "1")[:1] == "0" or \
                {f"sq_rq_mul_{n.bit_length() - 1}", f"sq2_block_mul_{n.bit_length() - 1}"} & set(B.kernel_timing_read())
            B.kernel_timing_enable(False)
            assert np.array_equal(_u64(c), wc) and np.array_equal(_u64(ce), wce), (a_ev, b_ev)
            assert np.array_equal(_u64(ae), wae) and np.array_equal(_u64(be), wbe), (a_ev, b_ev)
            c2 = torch.zeros_like(c)                            # no evals outputs: the Montgomery form of the product
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c2.data_ptr(), batch, a_is_evals=a_ev, b_is_evals=b_ev)
            assert torch.equal(c2, c), (a_ev, b_ev)
    # aliasing the reference's mul_mut / in-place uses produce: c over a; c over an evals input; a's evals over a
    da, db = dev(a), dev(b)
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), da.data_ptr(), batch)
    assert np.array_equal(_u64(da), wc)
    da, db = dev(wae), dev(b)
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), da.data_ptr(), batch, a_is_evals=True)
    assert np.array_equal(_u64(da), wc)
    da, db, c = dev(a), dev(b), torch.zeros((batch, n), dtype=torch.int64, device="cuda")
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c.data_ptr(), batch, d_a_evals=da.data_ptr())
    assert np.array_equal(_u64(da), wae) and np.array_equal(_u64(c), wc)


@pytest.mark.parametrize("q,n,batch", [(Q16, 32768, 3), (786433, 65536, 2), (786433, 131072, 1)])
def test_small_modulus_two_pass_transforms(pkg, oracle, q, n, batch):
    """2^15 <= n <= 2^17 at q < 2^32 / 25 (q = 65537's largest n; 786433 = 3 * 2^18 + 1): strided 32-bit pass + 2^14-point
    blocks with a u32 intermediate (smallq.hip).  Forward words == the oracle's, round trip == identity, in place too; the
    plain product is strided(a), strided(b), one middle kernel per block, strided inverse; the cached-evals product (61-bit
    kernels) consumes these transforms and agrees."""
    import torch

    B = pkg.binding
    plan = pkg.Plan(q, n)
    rng = np.random.default_rng(q + n)
    a = rng.integers(0, q, (batch, n), dtype=np.uint64)
    b = rng.integers(0, q, (batch, n), dtype=np.uint64)
    dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    da, db = dev(a), dev(b)
    A, r = torch.empty_like(da), torch.empty_like(da)
    B.kernel_timing_reset(); B.kernel_timing_enable(True)
    plan.forward_dev(da.data_ptr(), A.data_ptr(), batch)
    plan.inverse_dev(A.data_ptr(), r.data_ptr(), batch)
    torch.cuda.synchronize()
    names = set(B.kernel_timing_read())
    B.kernel_timing_enable(False)
    lg = n.bit_length() - 1
    ext32 = os.environ.get("FHE_EXT32", "1")[:1] != "0"
    assert not ext32 or {f"sq2_strided_fwd_{lg}", f"sq2_block_fwd_{lg}", f"sq2_block_inv_{lg}", f"sq2_strided_inv_{lg}"} <= names, names
    assert np.array_equal(_u64(A), oracle.ntt(q, n, a).reshape(batch, n))
    assert np.array_equal(_u64(r), a)
    x = da.clone()
    plan.forward_dev(x.data_ptr(), x.data_ptr(), batch)                  # in place
    assert torch.equal(x, A)
    plan.inverse_dev(x.data_ptr(), x.data_ptr(), batch)
    assert torch.equal(x, da)
    c, Bv, c2 = torch.empty_like(da), torch.empty_like(db), torch.empty_like(da)
    B.kernel_timing_reset(); B.kernel_timing_enable(True)
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c.data_ptr(), batch)
    torch.cuda.synchronize()
    assert not ext32 or f"sq2_block_mul_{lg}" in set(B.kernel_timing_read())
    B.kernel_timing_enable(False)
    assert np.array_equal(_u64(c), oracle.rq_mul(q, n, a, b)[0].reshape(batch, n))
    plan.forward_dev(db.data_ptr(), Bv.data_ptr(), batch)
    plan.rq_mul_dev(A.data_ptr(), Bv.data_ptr(), c2.data_ptr(), batch, a_is_evals=True, b_is_evals=True)
    assert torch.equal(c2, c)
    work = torch.empty(pkg.load_library().fhe_rq_mul_workspace_bytes(plan.handle, batch) // 8, dtype=torch.int64, device="cuda")
    c2.zero_()
    plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), c2.data_ptr(), batch, d_work=work.data_ptr())     # caller-owned scratch
    assert torch.equal(c2, c)


def test_small_prime_products_on_random_shapes(pkg, oracle):
    """A seeded sweep over the shapes digit32.hip serves (k = 1; n = 2^8 .. 2^12; any 1 <= l <= 64; batches that do and do
    not fill whole steps / parts / tail workgroups): external product and key switch, plain and prepared keys, word for
    word against the oracle."""
    import torch

    L, B = pkg.load_library(), pkg.binding
    rng = np.random.default_rng(0xD16175)
    k = 1
    for case in range(10):
        n = 1 << int(rng.integers(8, 13))
        l = int(rng.integers(1, 65))
        batch = int(rng.integers(1, 12)) if n >= 2048 else int(rng.integers(1, 40))
        # external product (full-range torus words)
        if n <= 1024 or case % 2 == 0:                      # the schoolbook oracle is O(n^2 l) per ciphertext
            nb = min(batch, 2 if n > 1024 else 6)
            g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
            c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (nb, k + 1, n), dtype=np.int64)).cuda()
            out = torch.empty_like(c)
            B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), out.data_ptr(), nb, None))
            assert np.array_equal(_u64(out), oracle.external_product(n, k, l, _u64(g), _u64(c))), ("ext", n, l, nb)
        # key switch (q61, or q = 65537 where every digit above bit 16 saturates)
        q = Q16 if case % 3 == 0 else Q61
        plan = pkg.Plan(q, n)
        glwe = rng.integers(0, q, (batch, k + 1, n), dtype=np.uint64)
        ksk = rng.integers(0, q, (k, l, k + 1, n), dtype=np.uint64)
        dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
        dglwe, dksk = dev(glwe), dev(ksk)
        dout = torch.empty_like(dglwe)
        if L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, dglwe.data_ptr(), dksk.data_ptr(), dout.data_ptr(), batch, 0, None) != 0:
            assert q == Q16 and l > 16                       # Zq::decompose is undefined there (q / 2^l = 0): rejected, as documented
            continue
        want = np.empty((batch, k + 1, n), dtype=np.uint64)
        for i in range(batch):
            oracle.glue("key_switch", q, n, k, 2, l, glwe[i], ksk, want[i])
        assert np.array_equal(_u64(dout), want), ("ks", q, n, l, batch)
        prep = torch.empty(L.fhe_glwe_ksk_prepared_words(plan.handle, k, 2, l), dtype=torch.int64, device="cuda")
        B._check(L.fhe_glwe_ksk_prepare_dev(plan.handle, k, 2, l, dksk.data_ptr(), prep.data_ptr(), None))
        dout.zero_()
        B._check(L.fhe_glwe_key_switch_prepared_dev(plan.handle, k, 2, l, dglwe.data_ptr(), prep.data_ptr(), dout.data_ptr(), batch, None))
        assert np.array_equal(_u64(dout), want), ("ks prepared", q, n, l, batch)


def test_fused_and_unfused_digit_paths_agree_at_bench_sizes(pkg):
    """FHE_DIGIT_MAC_FUSED=0 keeps round 1's materialised digit transforms + mac_rows_kernel; both forms
    must produce identical words for 630 external products (N=1024) and 256 key switches (N=4096)."""
    code = (
        "import sys, hashlib, numpy as np, torch; sys.path.insert(0, %r)\n"
        "import fhe_study_amd as pkg\n"
        "L, B = pkg.load_library(), pkg.binding\n"
        "rng = np.random.default_rng(1)\n"
        "n, k, l, batch = 1024, 1, 64, 630\n"
        "g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()\n"
        "c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()\n"
        "o = torch.empty_like(c)\n"
        "B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), o.data_ptr(), batch, None))\n"
        "print('ext', hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())\n"
        "q, n, k, l, batch = pkg.Q61, 4096, 1, 61, 256\n"
        "plan = pkg.Plan(q, n)\n"
        "glwe = torch.from_numpy(rng.integers(0, q, (batch, k + 1, n), dtype=np.int64)).cuda()\n"
        "ksk = torch.from_numpy(rng.integers(0, q, (k, l, k + 1, n), dtype=np.int64)).cuda()\n"
        "o = torch.empty_like(glwe)\n"
        "B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, glwe.data_ptr(), ksk.data_ptr(), o.data_ptr(), batch, 0, None))\n"
        "print('ks', hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())\n" % ROOT)
    outs = []
    # FHE_EXT32=0: both forms of the 61-bit kernels (with it on, these shapes take the small-prime kernels: next test)
    for fused in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FHE_DIGIT_MAC_FUSED=fused, FHE_EXT32="0"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("ext", "ks"))])
    assert len(outs[0]) == 2 and outs[0] == outs[1]


def test_small_prime_and_61_bit_kernels_agree_at_bench_sizes(pkg):
    """FHE_EXT32=0 keeps the 61-bit kernels for the external product, key switching and BFV; the default takes the
    27-bit-prime kernels (digit32.hip, bfv32.hip) where the shape allows.  Both must produce identical words: 630 external
    products (N=1024), 256 key switches (N=4096) with the plain and the prepared key, 64 BFV products at N=8192 and 4096."""
    code = (
        "import sys, hashlib, numpy as np, torch; sys.path.insert(0, %r)\n"
        "import fhe_study_amd as pkg\n"
        "L, B = pkg.load_library(), pkg.binding\n"
        "rng = np.random.default_rng(11)\n"
        "H = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()\n"
        "n, k, l, batch = 1024, 1, 64, 630\n"
        "g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()\n"
        "c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()\n"
        "o = torch.empty_like(c)\n"
        "B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), o.data_ptr(), batch, None))\n"
        "print('ext', H(o))\n"
        "q, n, k, l, batch = pkg.Q61, 4096, 1, 61, 256\n"
        "plan = pkg.Plan(q, n)\n"
        "glwe = torch.from_numpy(rng.integers(0, q, (batch, k + 1, n), dtype=np.int64)).cuda()\n"
        "ksk = torch.from_numpy(rng.integers(0, q, (k, l, k + 1, n), dtype=np.int64)).cuda()\n"
        "o = torch.empty_like(glwe)\n"
        "B._check(L.fhe_glwe_key_switch_dev(plan.handle, k, 2, l, glwe.data_ptr(), ksk.data_ptr(), o.data_ptr(), batch, 0, None))\n"
        "print('ks', H(o))\n"
        "prep = torch.empty(L.fhe_glwe_ksk_prepared_words(plan.handle, k, 2, l), dtype=torch.int64, device='cuda')\n"
        "B._check(L.fhe_glwe_ksk_prepare_dev(plan.handle, k, 2, l, ksk.data_ptr(), prep.data_ptr(), None))\n"
        "o.zero_()\n"
        "B._check(L.fhe_glwe_key_switch_prepared_dev(plan.handle, k, 2, l, glwe.data_ptr(), prep.data_ptr(), o.data_ptr(), batch, None))\n"
        "print('ksp', H(o))\n"
        "for n in (8192, 4096):\n"
        "    q, t, batch = 65537, 2, 64\n"
        "    pq = q * q * q\n"
        "    ab = torch.from_numpy(rng.integers(0, q, (4, batch, n), dtype=np.int64)).cuda()\n"
        "    rlk = torch.from_numpy(rng.integers(0, pq, (2, n), dtype=np.int64)).cuda()\n"
        "    o = torch.empty((2, batch, n), dtype=torch.int64, device='cuda')\n"
        "    B._check(L.fhe_bfv_mul_dev(q, n, t, pq, rlk.data_ptr(), ab.data_ptr(), o.data_ptr(), batch, None))\n"
        "    print('bfv', n, H(o))\n" % ROOT)
    outs = []
    for on in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FHE_EXT32=on), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("ext", "ks", "bfv"))])
    assert len(outs[0]) == 5 and outs[0] == outs[1]
    assert outs[0][1].split()[1] == outs[0][2].split()[1]            # plain and prepared key: the same words


def test_two_devices_from_two_host_threads(pkg, oracle):
    """INTEGRATION.md §5 as a C caller would do it: one host thread per device, hipSetDevice, the block
    of fhe_shard_range, shared plan, per-device tables.  Needs two GPUs: skipped on a one-GPU box."""
    import threading

    import torch

    B = pkg.binding
    world = B.device_count()
    if world < 2:
        pytest.skip("needs >= 2 HIP devices (fhe_ntt_device_count() < 2)")
    world = min(world, 8)
    q, n, batch = Q61, 65536, 5 * world + 1
    plan = pkg.Plan(q, n)                                   # shared by every thread
    a = oracle.fill_synthetic(q, 0x5EED, 0, batch * n).reshape(batch, n)
    want = oracle.ntt(q, n, a).reshape(batch, n)
    got, errs = [None] * world, []

    def worker(r):
        try:
            torch.cuda.set_device(r)                        # hipSetDevice for this host thread
            b0, b1 = B.shard_range(batch, world, r)
            B._check(pkg.load_library().fhe_ntt_plan_prepare(plan.handle))
            x = torch.from_numpy(a[b0:b1].view(np.int64).copy()).to(f"cuda:{r}")
            y = torch.empty_like(x)
            plan.forward_dev(x.data_ptr(), y.data_ptr(), b1 - b0, torch.cuda.current_stream(r).cuda_stream)
            torch.cuda.synchronize(r)
            got[r] = _u64(y).reshape(b1 - b0, n)
        except Exception as e:                              # noqa: BLE001 - reported below
            errs.append((r, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert np.array_equal(np.concatenate(got), want)


def test_fused_external_product_large_batch_single_part(pkg, oracle):
    """A batch that fills the chip by itself is not split into parts (one workgroup per ciphertext walks all
    its digits; the tail kernel then reads the sums directly): first and last 24 products against the oracle."""
    import torch

    L, B = pkg.load_library(), pkg.binding
    n, k, l, batch = 256, 1, 8, 2100
    rng = np.random.default_rng(2100)
    g = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (k + 1, l, k + 1, n), dtype=np.int64)).cuda()
    c = torch.from_numpy(rng.integers(-(1 << 63), 1 << 63, (batch, k + 1, n), dtype=np.int64)).cuda()
    out = torch.empty_like(c)
    B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.data_ptr(), out.data_ptr(), batch, None))
    rows = list(range(24)) + list(range(batch - 24, batch))
    want = oracle.external_product(n, k, l, _u64(g), _u64(c[rows]))
    assert np.array_equal(_u64(out[rows]), want)
    # every product is the same function of its ciphertext: reversing the batch reverses the outputs
    out2 = torch.empty_like(c)
    B._check(L.fhe_tggsw_external_product_dev(n, k, l, g.data_ptr(), c.flip(0).contiguous().data_ptr(), out2.data_ptr(), batch, None))
    assert torch.equal(out2.flip(0), out)


@pytest.mark.parametrize("q,n,p,t,batch", [(Q16, 16, Q16 * Q16, 2, 5), (Q16, 8192, Q16 * Q16, 2, 3), (Q61, 64, 1 << 1, 4, 2),
                                           (Q16, 16384, Q16 * Q16, 2, 1),     # 2n = 2^15: 7 strided stages in the fused last pass
                                           (12289, 32768, 12289, 3, 1)])      # 2n = 2^16: 8 strided stages; another modulus / t
def test_bfv_multiply_with_a_resident_relinearisation_key(pkg, oracle, q, n, p, t, batch):
    """fhe_bfv_rlk_prepare_dev once, then fhe_bfv_mul_prepared_dev / fhe_bfv_relinearize_prepared_dev: the words of
    the plain entry points and of the oracle's schoolbook (bfv/src/lib.rs:59-90,251-271), for several batches."""
    import torch

    L, B = pkg.load_library(), pkg.binding
    pq = p * q
    rng = np.random.default_rng(n + batch)
    ab = rng.integers(0, q, (4, batch, n), dtype=np.uint64)
    rlk = rng.integers(0, pq, (2, n), dtype=np.uint64)
    dev = lambda x: torch.from_numpy(x.view(np.int64).copy()).cuda()
    dab, drlk = dev(ab), dev(rlk)
    words = L.fhe_bfv_rlk_prepared_words(q, n, pq)
    assert words in (4 * n, 6 * n, 8 * n, 12 * n)       # opaque: 61-bit CRT primes (one, split key, or several) or three 27-bit primes
    prep = torch.empty(words, dtype=torch.int64, device="cuda")
    B._check(L.fhe_bfv_rlk_prepare_dev(q, n, pq, drlk.data_ptr(), prep.data_ptr(), None))
    plain = torch.empty((2, batch, n), dtype=torch.int64, device="cuda")
    B._check(L.fhe_bfv_mul_dev(q, n, t, pq, drlk.data_ptr(), dab.data_ptr(), plain.data_ptr(), batch, None))
    drlk.zero_()                                            # the prepared key stands alone
    for nb in (batch, 1):
        sub = dev(ab[:, :nb].copy())
        out = torch.empty((2, nb, n), dtype=torch.int64, device="cuda")
        B._check(L.fhe_bfv_mul_prepared_dev(q, n, t, pq, prep.data_ptr(), sub.data_ptr(), out.data_ptr(), nb, None))
        assert torch.equal(out, plain[:, :nb])
    w0, w1 = oracle.bfv_mul(q, n, t, pq, rlk[0], rlk[1], ab[0, :1], ab[1, :1], ab[2, :1], ab[3, :1])
    assert np.array_equal(_u64(plain[0, 0]), w0[0]) and np.array_equal(_u64(plain[1, 0]), w1[0])
    assert L.fhe_bfv_rlk_prepared_words(q, 3, pq) == 0
