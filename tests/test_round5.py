"""Round 5: `bench.py --gpus N` starts its own ranks when no launcher did, and the
transform kernels after their address arithmetic moved onto the scalar unit (NTT::ntt / intt, arith/src/ntt.rs:44-110):
every batch shape that takes the ragged (per-lane) load / store path and every one that takes the uniform path, word for
word against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import Q16, Q61, ROOT


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """a launcher that started a different number of ranks than --gpus is an error (exit 2), decided before torch or the
    GPU is touched — never a line whose n_gpus differs from what was asked"""
    env = _clean_env()
    env.update(RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 2, r.stderr[-2000:]
    assert "WORLD_SIZE=3" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 2


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_two_ranks(pkg):
    """plain `python bench.py --gpus 2 ...` with a clean environment: the process starts two ranks itself (before any
    HIP call of its own) and relays rank 0's line, which must say n_gpus 2; 513 rows = a ragged split (257 + 256)"""
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu",
           "--backend", "gloo", "--global-batch", "513", "--no-cpu-baseline", "--parity-all-ranks"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 513
    assert out["parity"]["mismatching_rows"] == 0 and out["parity"]["ranks_checked"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("q,log_n", [(Q61, 4), (Q61, 6), (Q61, 8), (Q61, 9), (Q61, 12), (Q61, 13), (Q61, 14), (Q61, 16), (Q61, 17),
                                     (Q16, 8), (Q16, 14), (4611686018425815041, 10), (4611686018425815041, 15),
                                     (0x1ffffff900000001, 12), (0x1ffffff900000001, 16),
                                     (9223372036844421121, 8), (9223372036844421121, 13), (4611686018429485057, 14),
                                     (9223372036844421121, 16), (4611686018429485057, 18)])
def test_every_group_shape_full_and_ragged(pkg, oracle, q, log_n):
    """a workgroup of the contiguous kernels holds W = 16 .. 1 polynomials: batches that fill every group (the uniform
    load / store path: one lane offset, scalar bases) and batches that leave the last group ragged (the per-lane path),
    forward, inverse and the product, on every arithmetic (pseudo-Mersenne, Shoup below and above 2^61, word Montgomery,
    the 32-bit kernels, and — since round 5 in the same kernels — the strict arithmetic of 2^62 <= q < 2^63)"""
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    n = 1 << log_n
    plan = pkg.Plan(q, n)
    for batch in ((1, 15, 16, 17, 32, 47) if log_n <= 12 else (1, 3, 16, 17)):
        a = oracle.fill_synthetic(q, 77 + batch + log_n, 0, batch * n).reshape(batch, n)
        a[0, :] = q - 1
        if batch > 1:
            a[batch - 1, ::2] = 0
        A = plan.forward(a)
        assert np.array_equal(np.asarray(A).reshape(-1), oracle.ntt(q, n, a).reshape(-1)), (q, n, batch)
        assert np.array_equal(np.asarray(plan.inverse(A)).reshape(-1), a.reshape(-1)), (q, n, batch)
        if batch in (1, 16, 17):
            b = oracle.fill_synthetic(q, 99 + batch + log_n, 0, batch * n).reshape(batch, n)
            got = plan.rq_mul(a, b)
            want = oracle.rq_mul(q, n, a, b)
            for x, y in zip(got, want):
                assert np.array_equal(np.asarray(x).reshape(-1), y.reshape(-1)), (q, n, batch)


@pytest.mark.gpu
@pytest.mark.parametrize("setting", [("B", 1, 1, 1), ("B", 1, 0, 2), ("E", 1, 1, 2), ("E", 1, 0, 4), ("D", 1, 1, 1)],
                         ids=lambda s: f"{s[0]}:{s[1]},{s[2]},{s[3]}")
def test_persistent_under_uneven_load(pkg, oracle, setting):
    """include/fhe_ntt_experimental.h: the one-launch kernels hand their intermediate over with plain stores, an agent-scope
    counter and 8-byte sc1 loads — a form the microarchitecture guide has no measured row for.  Its advice for any hand-off:
    test under UNEVEN load with an L1-warm consumer, every word compared.  Here a copy kernel streams on a second stream
    while the persistent kernel runs (its workgroups progress at different rates and are evicted and resumed unevenly),
    the rings are as short as the protocol allows (ring slots are rewritten and re-read hundreds of times by the same
    CUs: consumers meet lines they have read before), and all 300 x 65536 words must be the two-pass kernels' words, three
    launches in a row."""
    import torch

    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    B = pkg.binding
    n, batch = 1 << 16, 300
    plan = pkg.Plan(Q61, n)
    main = torch.cuda.current_stream()
    x = torch.empty(batch * n, dtype=torch.int64, device="cuda:0")
    want = torch.empty_like(x)
    got = torch.empty_like(x)
    B.fill_synthetic_dev(Q61, 0xF4E50555, 0, batch * n, x.data_ptr(), main.cuda_stream)
    try:
        B.set_persist(0)
        plan.forward_dev(x.data_ptr(), want.data_ptr(), batch, main.cuda_stream)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        src = torch.empty(96 << 20, dtype=torch.int64, device="cuda:0")        # 768 MiB: beyond the Infinity Cache
        dst = torch.empty_like(src)
        B.set_persist(*setting)
        for rep in range(3):
            got.zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(6):                                               # ~1.5 ms of streaming per copy
                    dst.copy_(src, non_blocking=True)
            plan.forward_dev(x.data_ptr(), got.data_ptr(), batch, main.cuda_stream)
            torch.cuda.synchronize()
            B.persist_status()
            assert torch.equal(got, want), (setting, rep, int((got != want).sum()))
    finally:
        B.set_persist(0)
    k = 2
    assert np.array_equal(want[: k * n].cpu().numpy().view(np.uint64), oracle.ntt(Q61, n, x[: k * n].cpu().numpy().view(np.uint64)).reshape(-1))


@pytest.mark.gpu
def test_short_lived_threads_do_not_accumulate_workspaces(pkg, oracle):
    """library workspaces are keyed by the calling thread; a pool of short-lived threads on one stream must not
    leave one buffer per thread that ever lived.  Forty threads, one after the other, each run a two-pass product (library
    workspace, d_work = NULL) on the NULL stream and exit: the bytes the library holds after the last one are what it held
    after the first (a dead thread's buffer is adopted by the next thread on that stream, in stream order), and every
    product is the oracle's."""
    import threading

    import torch

    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    L = pkg.load_library()
    q, n, batch = Q61, 16384, 5
    plan = pkg.Plan(q, n)
    a = oracle.fill_synthetic(q, 1700, 0, batch * n)
    b = oracle.fill_synthetic(q, 1800, 0, batch * n)
    want = oracle.rq_mul(q, n, a, b)[0]
    da = torch.from_numpy(a.view(np.int64).copy()).cuda()
    db = torch.from_numpy(b.view(np.int64).copy()).cuda()
    outs, held, errs = [], [], []

    def work():
        try:
            dc = torch.empty_like(da)
            plan.rq_mul_dev(da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, stream=None)
            outs.append(dc)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    for _ in range(40):
        t = threading.Thread(target=work)
        t.start()
        t.join()
        held.append(int(L.fhe_ntt_workspace_bytes()))
    torch.cuda.synchronize()
    assert not errs, errs
    assert held[0] > 0 and held[-1] == held[0], held
    for dc in outs:
        assert np.array_equal(dc.cpu().numpy().view(np.uint64), want.reshape(-1))


@pytest.mark.gpu
def test_shard_gather_from_the_c_abi(pkg, oracle):
    """fhe_shard_gather_dev: the optional gather of a block-partitioned batch for a caller without
    torch / RCCL.  Three 'ranks' (all on device 0 here: a 1-GPU box) each transform their fhe_shard_range block of a
    ragged batch in their own buffer; the gather puts them, in rank order, where the single-rank transform of the whole
    batch puts its rows.  An empty shard (more ranks than rows) may be NULL."""
    import torch

    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    B = pkg.binding
    q, n, total, world = Q61, 4096, 37, 3
    plan = pkg.Plan(q, n)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.empty(total * n, dtype=torch.int64, device="cuda:0")
    B.fill_synthetic_dev(q, 0xF4E50777, 0, total * n, x.data_ptr(), st)
    whole = torch.empty_like(x)
    plan.forward_dev(x.data_ptr(), whole.data_ptr(), total, st)
    shards = []
    for r in range(world):
        b, e = B.shard_range(total, world, r)
        out = torch.empty((e - b) * n, dtype=torch.int64, device="cuda:0")
        plan.forward_dev(x.data_ptr() + b * n * 8, out.data_ptr(), e - b, st)
        shards.append(out)
    gathered = torch.zeros_like(x)
    B.shard_gather_dev(total, n, [0] * world, [s.data_ptr() for s in shards], 0, gathered.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(gathered, whole)
    # more ranks than rows: the trailing shards are empty and may be NULL
    tiny = torch.zeros(2 * n, dtype=torch.int64, device="cuda:0")
    B.shard_gather_dev(2, n, [0] * 4, [whole.data_ptr(), whole.data_ptr() + n * 8, None, None], 0, tiny.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(tiny, whole[: 2 * n])
    assert np.array_equal(whole[:n].cpu().numpy().view(np.uint64), oracle.ntt(q, n, x[:n].cpu().numpy().view(np.uint64)).reshape(-1))


_G63_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
import fhe_study_amd as pkg
h = hashlib.sha256()
for q in (9223372036752015361, 4611686018429485057):       # the largest / smallest prime = 1 (mod 2^21) in [2^62, 2^63)
    for log_n, batch in ((4, 33), (9, 17), (13, 3), (14, 3), (16, 2), (18, 1), (20, 1)):
        n = 1 << log_n
        plan = pkg.Plan(q, n)
        rng = np.random.default_rng(q %% 1000 + log_n)
        a = rng.integers(0, q, size=(batch, n), dtype=np.uint64); a[0, :] = q - 1
        b = rng.integers(0, q, size=(batch, n), dtype=np.uint64); b[-1, ::3] = q - 1
        A = plan.forward(a)
        h.update(np.ascontiguousarray(A).tobytes())
        h.update(np.ascontiguousarray(plan.inverse(A)).tobytes())
        for x in plan.rq_mul(a, b):
            h.update(np.ascontiguousarray(x).tobytes())
print("DIGEST", h.hexdigest())
"""


@pytest.mark.gpu
def test_strict_moduli_two_builds_of_the_same_words(pkg):
    """2^62 <= q < 2^63 (NTT::ntt / intt / Rq x Rq, arith/src/ntt.rs:44-110, ring_nq.rs:586-607; Zq's limit zq.rs:225): since
    round 5 the range runs in the two-pass / fused kernels with strict butterflies (AR = 3) instead of ceil(log2 n / 4) plain
    launches.  The plain kernels are still in the library (n < 16, and everything under FHE_G63_PLAIN=1): both forms must
    produce the same words — forward, inverse and product (c, c_evals, a_evals, b_evals), single-pass and two-pass sizes,
    ragged batches.  (Each form against the oracle: test_moduli_between_2_62_and_2_63, test_every_group_shape_full_and_ragged.)"""
    assert pkg.binding.device_count() >= 1, "no HIP device: -m gpu tests need a real MI355X"
    script = _G63_SCRIPT % (ROOT,)
    digests = []
    for plain in ("0", "1"):
        env = _clean_env()
        env["FHE_G63_PLAIN"] = plain
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert digests[0] == digests[1], digests

