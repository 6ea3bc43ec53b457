"""Independent GPs / hyper-parameter restarts across ranks.

The reference's only data-parallel strategy is task parallelism over whole GPs: restarts under
`tools::par::max` (src/limbo/opt/parallel_repeater.hpp:86-105, src/limbo/tools/parallel.hpp:169-191)
and output dimensions under `tools::par::loop` (src/limbo/model/multi_gp.hpp:124-126).  Across the
8 GPUs of a node that is one process per GPU, each evaluating its own shard of the G units with no
data-path collective; the only exchange is the final arg-max, an all-gather of (value, theta)
records of a few hundred bytes (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard(n_units: int, rank: int, world: int) -> range:
    """Unit g runs on rank g mod world (multi_gp.hpp's loop index -> GPU)."""
    return range(rank, n_units, world)


def argmax_over_ranks(values, thetas, dist=None, device="cpu"):
    """values: this rank's objective values (len k); thetas: (k, T).  Returns (best_value,
    best_theta, owner_rank) identical on every rank — the parallel_reduce max of
    tools/parallel.hpp:169-191.  Ranks may hold different numbers of units (ragged shards)."""
    import torch

    values = np.atleast_1d(np.asarray(values, dtype=np.float64))
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    T = thetas.shape[1] if thetas.size else 0
    if len(values):
        i = int(np.argmax(values))
        rec = np.concatenate([[values[i]], thetas[i]])
    else:  # a rank without units never wins
        rec = np.concatenate([[-np.inf], np.zeros(T)])
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
