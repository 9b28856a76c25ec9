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
                                     (0x1ffffff900000001, 12), (0x1ffffff900000001, 16)])
def test_every_group_shape_full_and_ragged(pkg, oracle, q, log_n):
    """a workgroup of the contiguous kernels holds W = 16 .. 1 polynomials: batches that fill every group (the uniform
    load / store path: one lane offset, scalar bases) and batches that leave the last group ragged (the per-lane path),
    forward, inverse and the product, on every arithmetic (pseudo-Mersenne, Shoup below and above 2^61, word Montgomery,
    the 32-bit kernels)"""
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
