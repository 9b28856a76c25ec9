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
