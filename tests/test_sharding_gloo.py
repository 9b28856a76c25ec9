"""CPU, world_size 2, gloo: the N>1 path — block partition of the batch and the optional
all-gather of result shards.  No GPU here, so the per-rank engine is stood in for by the
CPU oracle (test infrastructure); what is under test is the sharding/gather plumbing that
bench.py and downstream callers use with the HIP engine on real ranks.  (The HIP engine itself
under a process group is driven by the -m gpu tests: tests/test_round2_gpu.py runs bench.py as two
ranks on one GPU, tests/test_round3_gpu.py runs it through RCCL and in its strong-scaling form.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import Q61, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, n, q, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fhe_study_amd as pkg
    from fhe_study_amd import sharding  # noqa: F401  (alias module resolves the package)
    from oracle import load_oracle

    O = load_oracle()

    def transform(rows):  # stand-in for Plan.forward_dev on this rank's block
        a = rows.numpy().view(np.uint64)
        return torch.from_numpy(O.ntt(q, n, a).reshape(a.shape).view(np.int64))

    def rows_fn(b0, b1):  # each rank generates its own rows, as bench.py does
        x = O.fill_synthetic(q, 0x5EED, b0 * n, (b1 - b0) * n)
        return torch.from_numpy(x.view(np.int64).reshape(b1 - b0, n))

    eng = pkg.sharding.ShardedNTT(transform)
    b0, b1 = pkg.sharding.shard_range(batch, world, rank)
    local = eng.forward_sharded(rows_fn, batch, gather=False)
    assert local.shape == (b1 - b0, n)
    full = eng.forward_sharded(rows_fn, batch, gather=True)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    # transform and gather overlapped piece by piece (async collectives): same batch
    for rows in (1, 2, 5):
        again = eng.forward_sharded(rows_fn, batch, gather=True, overlap_rows=rows)
        assert torch.equal(again, full), (rank, rows)
    # a transform that fills a caller's buffer (Plan.forward_dev's shape): the short last shard is written straight into
    # the equal-count send buffer
    def transform_out(rows, out=None):
        r = transform(rows)
        if out is None:
            return r
        out.copy_(r)
        return out

    eng_out = pkg.sharding.ShardedNTT(transform_out)
    assert eng_out._takes_out and not eng._takes_out
    assert torch.equal(eng_out.forward_sharded(rows_fn, batch, gather=True), full), rank
    assert torch.equal(eng_out.forward_sharded(rows_fn, batch, gather=True, overlap_rows=2), full), rank
    # the same gather in chunks of rows (several collectives + copies into place): same batch
    for chunk in (1, 2, 3, 64):
        again = pkg.sharding.all_gather_rows(local, batch, chunk_rows=chunk)
        assert torch.equal(again, full), (rank, chunk)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 5, 1])
def test_two_rank_shard_and_gather(tmp_path, oracle, batch):
    n, q, world = 256, Q61, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, batch, n, q, str(tmp_path)), nprocs=world, join=True)
    x = oracle.fill_synthetic(q, 0x5EED, 0, batch * n)
    want = oracle.ntt(q, n, x).reshape(batch, n).view(np.int64)
    for r in range(world):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert np.array_equal(got, want), f"rank {r}"


def test_shard_range_is_a_partition():
    import fhe_study_amd as pkg

    for batch in (0, 1, 7, 8, 630, 65536):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                b0, b1 = pkg.sharding.shard_range(batch, world, r)
                assert 0 <= b0 <= b1 <= batch
                rows += list(range(b0, b1))
            assert rows == list(range(batch))
    # BASELINE.json configs[3]: 630 bootstraps over 8 GPUs = 79 x 7 + 77
    assert [pkg.sharding.shard_range(630, 8, r)[1] - pkg.sharding.shard_range(630, 8, r)[0] for r in range(8)] == [79] * 7 + [77]
