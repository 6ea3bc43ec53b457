"""N > 1 path on CPU: two gloo ranks shard independent units and agree on the arg-max
(limbo_amd/parallel.py — the only collective of the design, bench.py uses the same function
over RCCL).  No GPU: the per-unit objective values are synthetic."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from limbo_amd import parallel as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
G = int(sys.argv[2])
rng = np.random.default_rng(7)
vals_all = rng.normal(size=G)            # same on every rank: the "objective" of unit g
th_all = rng.normal(size=(G, 5))
mine = list(P.shard(G, rank, world))
v, th, owner = P.argmax_over_ranks(vals_all[mine], th_all[mine], dist)
g = int(np.argmax(vals_all))
assert v == vals_all[g] and np.array_equal(th, th_all[g]) and owner == g % world, (rank, v, owner)
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": True, "best": g, "owner": owner, "units_rank0": len(mine)}))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, G):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, str(ROOT), str(G)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    return outs[0][0]


def test_two_ranks_agree_on_argmax():
    assert '"ok": true' in _run(2, 64)


def test_eight_ranks_agree_on_argmax():
    """the world size of the driver's scaling run (one rank per GPU of an 8-GPU node): 64 units = configs[3]'s 64 GPs dealt
    8 per rank, and 5 units over 8 ranks (three idle ranks)."""
    assert '"ok": true' in _run(8, 64)
    assert '"ok": true' in _run(8, 5)


def test_ragged_shards_and_idle_rank():
    # 3 units over 2 ranks (2 + 1) and 1 unit over 2 ranks (rank 1 idle)
    assert '"ok": true' in _run(2, 3)
    assert '"ok": true' in _run(2, 1)


def test_shard_covers_every_unit_once():
    from limbo_amd import parallel as P

    for world in (1, 2, 3, 8):
        seen = sorted(g for r in range(world) for g in P.shard(64, r, world))
        assert seen == list(range(64))


QUERY_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from limbo_amd import _capi, parallel as P
from oracle import np_oracle as O
from oracle import binding as OB
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
M = int(sys.argv[2])
rng = np.random.default_rng(3)
X = rng.uniform(0, 1, size=(60, 3)); Y = np.stack([np.sin(X.sum(1)), np.cos(X[:, 0])], axis=1)
om, _ = O.obs_mean_data(Y)
# every rank holds its own replica of the GP (here: the CPU oracle stands in for the engine)
h = _capi.Handle(OB.load_oracle()); h.set_data(X, om); h.set_kernel(O.MATERN52, np.array([0.1, 0.2]), 0.01)
assert h.compute() == 0
Xq = rng.uniform(0, 1, size=(M, 3))
kta, var = P.query_sharded(h.query_batch, Xq, dist)
k0, v0 = h.query_batch(Xq) if M else (np.zeros((0, 2)), np.zeros(0))
assert kta.shape == (M, 2) and var.shape == (M,)
assert np.array_equal(kta, np.asarray(k0).reshape(M, 2)) and np.array_equal(var, np.asarray(v0).reshape(M))
dist.barrier()
if rank == 0:
    print(json.dumps({"ok": True, "M": M}))
dist.destroy_process_group()
"""


def _run_query(world, M):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", QUERY_WORKER, str(ROOT), str(M)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    return outs[0][0]


def test_sharded_query_two_ranks():
    """config 3's query batch sharded over M: every rank answers a slice, one all-gather reassembles;
    identical to the unsharded answer, including ragged (odd M) and a rank with an empty slice (M = 1)."""
    assert '"ok": true' in _run_query(2, 101)
    assert '"ok": true' in _run_query(2, 1)


def test_row_slice_partitions():
    from limbo_amd import parallel as P

    for M in (0, 1, 7, 64, 101):
        for world in (1, 2, 3, 8):
            idx = [i for r in range(world) for i in range(M)[P.row_slice(M, r, world)]]
            assert idx == list(range(M))
