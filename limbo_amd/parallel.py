"""Independent GPs / hyper-parameter restarts across ranks.

The reference's only data-parallel strategy is task parallelism over whole GPs: restarts under
`tools::par::max` (src/limbo/opt/parallel_repeater.hpp:86-105, src/limbo/tools/parallel.hpp:169-191)
and output dimensions under `tools::par::loop` (src/limbo/model/multi_gp.hpp:124-126).  Across the
8 GPUs of a node that is one process per GPU, each evaluating its own shard of the G units with no
data-path collective; the only exchange is the final arg-max, an all-gather of (value, theta)
records of a few hundred bytes (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).

Batched queries shard over the M query points instead: every rank holds the
same GP (it factors its own copy — cheaper than broadcasting a 2 GiB factor), answers a contiguous
slice of the points, and one all-gather of (P + 1) doubles per point reassembles mu and sigma^2.
"""
from __future__ import annotations

import numpy as np


def shard(n_units: int, rank: int, world: int) -> range:
    """Unit g runs on rank g mod world (multi_gp.hpp's loop index -> GPU)."""
    return range(rank, n_units, world)


def argmax_over_ranks(values, thetas, dist=None, device="cpu", force=False):
    """values: this rank's objective values (len k); thetas: (k, T).  Returns (best_value,
    best_theta, owner_rank) identical on every rank — the parallel_reduce max of
    tools/parallel.hpp:169-191.  Ranks may hold different numbers of units (ragged shards).
    force: run the two collectives even in a group of one rank (bench.py --force-dist: the RCCL path
    rehearsed on a one-GPU box)."""
    import torch

    values = np.atleast_1d(np.asarray(values, dtype=np.float64))
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    T = thetas.shape[1] if thetas.size else 0
    if len(values):
        i = int(np.argmax(values))
        rec = np.concatenate([[values[i]], thetas[i]])
    else:  # a rank without units never wins
        rec = np.concatenate([[-np.inf], np.zeros(T)])
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(rec[0]), rec[1:].copy(), 0
    # T can differ only if a rank had no unit: agree on the record length first
    n = torch.tensor([len(rec)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    buf = np.full(int(n.item()), 0.0)
    buf[: len(rec)] = rec
    t = torch.tensor(buf, dtype=torch.float64, device=device)
    allrec = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allrec, t)
    allrec = torch.stack(allrec).cpu().numpy()
    owner = int(np.argmax(allrec[:, 0]))  # first maximum: deterministic tie-break by rank
    return float(allrec[owner, 0]), allrec[owner, 1:].copy(), owner


def row_slice(n_rows: int, rank: int, world: int) -> slice:
    """Contiguous, balanced slice of n_rows for this rank (the first n_rows % world ranks get one more)."""
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return slice(lo, lo + q + (1 if rank < r else 0))


def query_sharded(query_fn, Xq, dist=None, device="cpu", force=False, stats=None):
    """query_fn(points) -> (kta [m x P], var [m]) on this rank's replica of the GP (e.g.
    Handle.query_batch).  Returns the full (kta [M x P], var [M]) on every rank.
    force: run the collectives even in a group of one rank (bench.py --force-dist).
    stats (a dict, optional): this rank's seconds in its own slice ("local_s") and in the collectives ("gather_s")."""
    import time

    import torch

    Xq = np.ascontiguousarray(Xq, dtype=np.float64)
    M = Xq.shape[0]
    t0 = time.perf_counter()
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        kta, var = query_fn(Xq)
        if stats is not None:
            stats.update(local_s=time.perf_counter() - t0, gather_s=0.0, points_local=M)
        return np.asarray(kta, float).reshape(M, -1), np.asarray(var, float).reshape(M)
    rank, world = dist.get_rank(), dist.get_world_size()
    sl = row_slice(M, rank, world)
    m = sl.stop - sl.start
    if m > 0:
        kta, var = query_fn(Xq[sl])
        kta = np.asarray(kta, float).reshape(m, -1)
    else:  # more ranks than points: nothing to answer, P learnt from the others
        kta, var = np.zeros((0, 0)), np.zeros(0)
    t1 = time.perf_counter()
    P = torch.tensor([kta.shape[1]], dtype=torch.int64, device=device)
    dist.all_reduce(P, op=dist.ReduceOp.MAX)  # a rank with an empty slice does not know P
    P = int(P.item())
    rows = (M + world - 1) // world  # all_gather wants equal shapes: pad every slice to the longest
    buf = np.zeros((rows, P + 1))
    if m > 0:
        buf[:m, :P] = kta
        buf[:m, P] = np.asarray(var, float).reshape(-1)
    t = torch.tensor(buf, dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = np.concatenate([parts[r].cpu().numpy()[: row_slice(M, r, world).stop - row_slice(M, r, world).start]
                          for r in range(world)])
    if stats is not None:
        stats.update(local_s=t1 - t0, gather_s=time.perf_counter() - t1, points_local=m)
    return out[:, :P].copy(), out[:, P].copy()
