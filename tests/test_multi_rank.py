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


def test_ragged_shards_and_idle_rank():
    # 3 units over 2 ranks (2 + 1) and 1 unit over 2 ranks (rank 1 idle)
    assert '"ok": true' in _run(2, 3)
    assert '"ok": true' in _run(2, 1)


def test_shard_covers_every_unit_once():
    from limbo_amd import parallel as P

    for world in (1, 2, 3, 8):
        seen = sorted(g for r in range(world) for g in P.shard(64, r, world))
        assert seen == list(range(64))
