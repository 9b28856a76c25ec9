"""Batch sharding across the GPUs of one node.

The transform has no exchange step: polynomials are independent, so rank r owns the
contiguous block of rows [r*ceil(B/G), min(B, (r+1)*ceil(B/G))) and runs the
single-GPU engine on it (one process per GPU).  The only collective is the optional
all-gather of result shards for a consumer that needs the whole batch on every rank
(RCCL over xGMI: backend "nccl"; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(batch, world, rank):
    """Rows [b0, b1) of the global batch owned by `rank`."""
    per = -(-batch // world)
    b0 = min(batch, rank * per)
    return b0, min(batch, b0 + per)


def all_gather_rows(local_rows, batch, group=None, chunk_rows=0):
    """local_rows: (b1-b0, n) int64 tensor of this rank's shard → (batch, n) on every rank.
    chunk_rows = 0: ONE all_gather_into_tensor per call (xGMI links are per-peer: fewer, larger
    collectives).  A full shard (every rank when batch % world == 0 — config 5: 8192 rows, 4 GiB per
    rank) is sent as it is, no copy; only a rank whose block is short (the ragged tail: 77 of 79 rows
    at 630 over 8) stages its rows in a ceil(B/G)-row buffer, because the collective needs equal counts.
    chunk_rows = R > 0: the shard travels in ceil(per / R) collectives of R rows per rank, each landing
    in a (world, R, n) staging buffer that is then copied into place — for a consumer that starts on
    the first rows while the rest is still on the links, and to bound the collective's size."""
    world = dist.get_world_size(group)
    per = -(-batch // world)
    n = local_rows.shape[1]
    send = local_rows
    if local_rows.shape[0] != per:
        send = torch.zeros((per, n), dtype=local_rows.dtype, device=local_rows.device)
        send[: local_rows.shape[0]] = local_rows
    send = send.contiguous()
    out = torch.empty((world * per, n), dtype=local_rows.dtype, device=local_rows.device)
    if not chunk_rows or chunk_rows >= per:
        dist.all_gather_into_tensor(out, send, group=group)
        return out[:batch]
    placed = out.view(world, per, n)
    stage = torch.empty((world * chunk_rows, n), dtype=local_rows.dtype, device=local_rows.device)
    for c0 in range(0, per, chunk_rows):
        rows = min(per, c0 + chunk_rows) - c0
        st = stage[: world * rows]                      # (world * rows, n): the flat form every backend accepts
        dist.all_gather_into_tensor(st, send[c0:c0 + rows], group=group)
        placed[:, c0:c0 + rows] = st.view(world, rows, n)
    return out[:batch]


class ShardedNTT:
    """Per-rank driver: `transform(rows) -> rows` is the single-device engine applied to
    this rank's block (on a GPU box: Plan.forward_dev on device tensors).

    Stream contract: the collectives are issued on torch's CURRENT stream.  A `transform` that enqueues its kernels
    elsewhere (Plan.forward_dev takes any stream) must say so — `stream=` a torch.cuda.Stream / ExternalStream: after
    every transform an event is recorded there and the current stream waits for it before the rows go on the links —
    or synchronise by itself.  With the default (None) the transform is taken to run on the current stream.
    `transform` may accept `out=` (a preallocated (rows, n) tensor to fill): the ragged last shard is then written
    straight into the equal-count send buffer instead of being copied into it."""

    def __init__(self, transform, group=None, stream=None):
        self.transform = transform
        self.group = group
        self.stream = stream
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        import inspect

        try:
            self._takes_out = "out" in inspect.signature(transform).parameters
        except (TypeError, ValueError):
            self._takes_out = False

    def _ordered(self, t):
        """make torch's current stream wait for the transform's stream (no-op for CPU tensors / the same stream)"""
        if self.stream is not None and t is not None and t.is_cuda:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
        return t

    def forward_sharded(self, full_batch_rows_fn, batch, gather=False, overlap_rows=0):
        """full_batch_rows_fn(b0, b1) materialises this rank's input rows (inputs are
        generated / loaded on the owning rank, never broadcast).
        overlap_rows = R > 0 (with gather): the shard is transformed R rows at a time and each finished piece is put on the
        links at once (async all-gather) while the next piece is transformed — the collective is link-bound and an order
        of magnitude longer than the transform, so a consumer of the gathered batch waits for the links only.  Two send
        and two staging buffers are reused in turn (a piece is copied into place as soon as its collective has completed),
        so the extra memory is four pieces, not twice the gathered batch."""
        b0, b1 = shard_range(batch, self.world, self.rank)
        per = -(-batch // self.world)                      # rows every rank sends (the ragged tail is padded)
        if not (gather and overlap_rows):
            rows = full_batch_rows_fn(b0, b1)
            if gather and self._takes_out and b1 - b0 != per and b1 > b0:
                # the short last shard: transformed straight into the equal-count send buffer (no padded copy afterwards)
                send = torch.empty((per, rows.shape[1]), dtype=rows.dtype, device=rows.device)
                send[b1 - b0:].zero_()
                self._ordered(self.transform(rows, out=send[: b1 - b0]))
                return all_gather_rows(send, batch, self.group)
            local = self._ordered(self.transform(rows))
            return all_gather_rows(local, batch, self.group) if gather else local
        out, bufs, inflight = None, [], []

        def land(item):
            work, stage, c0, c1 = item
            work.wait()
            out[:, c0:c1] = stage.view(self.world, c1 - c0, out.shape[2])

        for i, c0 in enumerate(range(0, per, overlap_rows)):
            c1 = min(per, c0 + overlap_rows)
            r0, r1 = min(b1, b0 + c0), min(b1, b0 + c1)     # this rank's real rows of the piece (may be fewer, or none)
            piece = self._ordered(self.transform(full_batch_rows_fn(r0, r1))) if r1 > r0 else None
            if out is None:
                ref = piece if piece is not None else self.transform(full_batch_rows_fn(b0, min(b1, b0 + 1)))
                n, dt, dev = ref.shape[1], ref.dtype, ref.device
                out = torch.empty((self.world, per, n), dtype=dt, device=dev)
                bufs = [(torch.empty((overlap_rows, n), dtype=dt, device=dev),
                         torch.empty((self.world * overlap_rows, n), dtype=dt, device=dev)) for _ in range(2)]
            if len(inflight) == 2:                          # the buffers of two pieces ago are needed again
                land(inflight.pop(0))
            send, stage = bufs[i % 2]
            send, stage = send[: c1 - c0], stage[: self.world * (c1 - c0)]
            if piece is not None:
                send[: r1 - r0] = piece
            send[r1 - r0:].zero_()
            inflight.append((dist.all_gather_into_tensor(stage, send, group=self.group, async_op=True), stage, c0, c1))
        for item in inflight:
            land(item)
        return out.view(self.world * per, n)[:batch]
