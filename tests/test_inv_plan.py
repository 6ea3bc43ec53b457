"""The launch plan of the recursive K^-1 (limbo_amd/csrc/inv2.hip; GP::compute_inv_kernel, src/limbo/model/gp.hpp:254-264)
EXECUTED IN NUMPY — host logic, no device.  gpe_debug_inv_plan hands out every tile product of the plan as
(buffer, offset, depth, valid rows, valid columns, chunk number, transposed destination) rows in launch order, share by share;
here each is carried out literally on column-major numpy buffers, with the leaf inverses (what k_inv_panels leaves) from numpy:
U must come out as L^-T, the lower triangle of K^-1 as LAPACK's — for any order N, ragged ones included (the last panel and the
last tiles are partial: a tile stores its valid part only, k ranges run to N rounded up to 64 over the zero-filled pads of U and
of the T-form / W buffer).  Round 6: no fold launches — chunk 0 of a cut k range goes to the tile, chunk c to partial
buffer c, and the launch itself adds them up in their order (the workgroup that finishes last does) and writes the transposed
copy.  Also held: the k range of no tile is cut into more than 1 + 3 chunks; every chunk of a cut tile names the same tile, counter
word and transposed destination; every product launch is dealt into shares whose longest is
no longer than the mean share + one chunk; every tile a product reads was written before (nothing relies on zero-filled memory
inside the N x N part); and a launch never reads what it also writes."""
import ctypes

import numpy as np
import pytest

from limbo_amd import _capi

TILE = 128


def _plan(n, ld, nbins=512, load_pct=100):
    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_inv_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64]
    rows = f(n, ld, nbins, load_pct, None, 0)
    assert rows > 0
    out = np.zeros((rows, 16), dtype=np.int64)
    assert f(n, ld, nbins, load_pct, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), rows) == rows
    return out


def _tile(buf, off, ld, rows, cols):
    """view of the rows x cols block whose origin is at offset `off` of a column-major buffer with leading dimension ld"""
    r, c = off % ld, off // ld
    assert r + rows <= buf.shape[0] and c + cols <= buf.shape[1], "a tile reaches beyond its buffer"
    return buf[r:r + rows, c:c + cols]


def _run_plan(n, ld, nbins, load_pct, seed):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n)) / np.sqrt(n)
    Kmat = A @ A.T + 0.3 * np.eye(n)
    L = np.linalg.cholesky(Kmat)
    cap = (n + 63) // 64 * 64  # (engine.hip, alloc_dev: the capacity is a multiple of 64)
    NAN = np.nan
    bufs = [np.full((ld, cap), NAN) for _ in range(7)]  # [row, col] = offset row + col * ld
    bufs[0][:n, :n] = L
    for b in (1, 3):  # U and the T-form / W buffer: zero-filled beyond the N x N part (engine.hip, inv2_zero_pads)
        bufs[b][:, :] = 0.0
        bufs[b][:n, :n] = NAN
    # the leaves (k_inv_panels): X_p into the T-form buffer, X_p^T into U, full squares (the last one partial)
    for p0 in range(0, n, 256):
        X = np.linalg.inv(L[p0:p0 + 256, p0:p0 + 256])
        X = np.tril(X)
        pw = X.shape[0]
        bufs[3][p0:p0 + pw, p0:p0 + pw] = X
        bufs[1][p0:p0 + pw, p0:p0 + pw] = X.T
    plan = _plan(n, ld, nbins, load_pct)
    steps = plan[:, 0]
    stats = dict(launches=int(steps.max()) + 1, products=0, max_chunks=1, transposed=0)
    for s in range(int(steps.max()) + 1):
        rows = plan[steps == s]
        kind = rows[0, 1]
        assert (rows[:, 1] == kind).all() and kind in (0, 2)
        te = 64 if kind & 2 else 128  # tile edge of the launch
        stats["tile_edges"] = stats.get("tile_edges", set()) | {te}
        stats["products"] += len(rows)
        # what the launch reads and writes, tile by tile: disjoint
        reads = {(int(r[2]), int(r[3]) % ld // te, (int(r[3]) // ld + kk) // te) for r in rows for kk in range(0, int(r[8]), 64)}
        reads |= {(int(r[4]), int(r[5]) % ld // te, (int(r[5]) // ld + kk) // te) for r in rows for kk in range(0, int(r[8]), 64)}
        writes = {(int(r[6]), int(r[7]) % ld // te, int(r[7]) // ld // te) for r in rows}
        writes |= {(int(r[14]), int(r[15]) % ld // te, int(r[15]) // ld // te) for r in rows if r[14] >= 0}
        assert not (reads & writes), "a launch reads a tile it also writes"
        share = rows[:, 9] >> 1
        assert (np.diff(share) >= 0).all()  # (rows come share by share)
        cut = {}  # counter word -> [tile (buf, off), mr, nc, T, chunks seen]
        tiles_w = set()
        for r in rows:
            _, _, ab, ao, bb, bo, cb, co, k, neg, mr, nc, sq, slot, tb, to = (int(v) for v in r)
            neg &= 1
            seq, nch = sq & 255, sq >> 8
            assert k % 64 == 0 and k >= 64 and 1 <= mr <= te and 1 <= nc <= te and 0 <= seq < nch <= 4
            assert ao % ld + mr <= n and bo % ld + nc <= n and co % ld + mr <= n and co // ld + nc <= n  # valid parts lie inside N
            At, Bt = _tile(bufs[ab], ao, ld, mr, k), _tile(bufs[bb], bo, ld, nc, k)
            assert not np.isnan(At).any() and not np.isnan(Bt).any(), "a product reads memory nothing wrote"
            v = At @ Bt.T
            v = -v if neg else v
            stats["max_chunks"] = max(stats["max_chunks"], nch)
            assert (cb <= 3) == (seq == 0) and (seq == 0 or cb == 3 + seq), "chunk 0 to the tile, chunk c to partial buffer c"
            assert (cb, co) not in tiles_w, "two products of one launch write the same place"
            tiles_w.add((cb, co))
            _tile(bufs[cb], co, ld, mr, nc)[:, :] = v
            if nch > 1:
                e = cut.setdefault(slot, [None, mr, nc, (tb, to), set(), nch, co])
                assert e[1:4] == [mr, nc, (tb, to)] and e[5] == nch and e[6] == co and seq not in e[4], "the chunks of a tile disagree"
                e[4].add(seq)
                if seq == 0:
                    e[0] = (cb, co)
            elif tb >= 0:
                _tile(bufs[tb], to, ld, nc, mr)[:, :] = v.T
                stats["transposed"] += 1
        # ... and what the launch's last-to-count workgroups do: D + P1 + P2 + P3 in that order, then the transposed copy
        for slot, (tile, mr, nc, (tb, to), seen, nch, co) in cut.items():
            assert tile is not None and seen == set(range(nch)), "a cut tile is short of chunks"
            D = _tile(bufs[tile[0]], tile[1], ld, mr, nc)
            for c in range(1, nch):
                P = _tile(bufs[3 + c], co, ld, mr, nc)
                assert not np.isnan(P).any()
                D += P
            assert not np.isnan(D).any()
            if tb >= 0:
                _tile(bufs[tb], to, ld, nc, mr)[:, :] = D.T
                stats["transposed"] += 1
    for b in (1, 3):  # the pads are as they were: no launch wrote there
        Z = bufs[b].copy()
        Z[:n, :n] = 0.0
        assert not Z.any()
    U = bufs[1][:n, :n]
    Kinv = bufs[2][:n, :n]
    return L, Kmat, U, Kinv, stats, plan


@pytest.mark.parametrize("n,ld,nbins,load_pct", [(1024, 1024 + 32, 512, 100), (2048, 2048 + 32, 512, 100), (1536, 1536 + 16, 64, 100),
                                                  (1280, 1280 + 32, 16, 50), (768, 800, 8, 200), (512, 544, 512, 100), (256, 272, 4, 100),
                                                  (1536, 1536 + 32, -4, 100),  # what a batched sequence of >= 4 members runs — no k range is cut
                                                  (1100, 1152 + 32, 512, 100), (1700, 1728 + 16, 512, 100), (1025, 1088 + 32, 512, 100),
                                                  (1344, 1344 + 32, 64, 100), (2047, 2048 + 32, 512, 100), (1281, 1344 + 16, -8, 100),
                                                  (300, 320 + 16, 8, 100), (65, 128 + 16, 4, 100)])
def test_inv_plan_executed_in_numpy(n, ld, nbins, load_pct):
    L, Kmat, U, Kinv, stats, plan = _run_plan(n, ld, nbins, load_pct, seed=n + nbins)
    Uref = np.linalg.inv(L).T
    iu = np.triu_indices(n)
    assert not np.isnan(U[iu]).any()
    assert np.max(np.abs(U[iu] - Uref[iu])) <= 1e-10 * np.max(np.abs(Uref))
    il = np.tril_indices(n)
    Kref = np.linalg.inv(Kmat)
    assert not np.isnan(Kinv[il]).any()
    assert np.max(np.abs(Kinv[il] - Kref[il])) <= 1e-9 * np.max(np.abs(Kref))
    assert stats["max_chunks"] <= 4
    if nbins < 0:
        assert stats["max_chunks"] == 1 and (plan[:, 6] <= 3).all()  # a batch cuts no k range: no partial buffer is ever written
    # algorithmic flops: the products of the plan do 2 n^3 / 3 less what the leaves did, at tile granularity
    prods = plan
    flops = float(np.sum(np.where(prods[:, 1] & 2, 64.0 * 64.0, 128.0 * 128.0) * 2.0 * prods[:, 8]))
    nn = (n + 127) // 128 * 128  # (a ragged order pays for whole tiles)
    assert 0.55 * 2 * n ** 3 / 3 - 2.0 * 256 ** 3 <= flops <= 1.4 * 2 * nn ** 3 / 3 + 2.0 * 128 ** 3 * 3  # 2 n^3 / 3 less the leaves, whole tiles on the diagonals
    print(f"n={n}: {stats['launches']} launches after the leaves, {stats['products']} tile products, {stats['transposed']} transposed copies")


def test_inv_plan_of_a_large_batch_cuts_nothing():
    """the plan of a batch of >= 4 members used to switch chunking off through a magic load of 1e9, whose product with the
    mean share overflowed the chunk length's int from N ~ 32768 on (undefined behaviour).  Now an explicit flag: at N = 40000
    (planned, not executed) every tile product keeps its whole k range, no partial buffer appears, a share per product."""
    n = 40000
    ld = (n + 63) // 64 * 64 + 32
    plan = _plan(n, ld, -4, 100)
    prods = plan
    assert len(prods) > 10000 and (prods[:, 8] >= 64).all() and (prods[:, 8] % 64 == 0).all()
    assert (prods[:, 12] == 1 << 8).all() and (prods[:, 6] <= 3).all()  # every product one chunk, no partial buffers
    for s_ in np.unique(prods[:, 0]):
        rows = prods[prods[:, 0] == s_]
        assert len(np.unique(rows[:, 9] >> 1)) == len(rows)  # every product a workgroup of its own
    # the largest product of the top level reads half the matrix's depth
    assert prods[:, 8].max() >= n // 2 - 256


def test_inv_plan_shares_are_balanced():
    """N = 4096 as config 2 runs it (512 shares = two resident workgroups per CU): per product launch that fills the chip the
    longest share is within 15 % + one chunk of the mean share."""
    n, ld = 4096, 4096 + 32
    plan = _plan(n, ld, 512, 100)
    steps = plan[:, 0]
    launches, big = 0, 0
    for s in range(int(steps.max()) + 1):
        rows = plan[steps == s]
        launches += 1
        units = rows[:, 8] // (64 if rows[0, 1] & 2 else 128)
        share = rows[:, 9] >> 1
        assert share.max() < 512
        if len(rows) < 256:
            continue
        big += 1
        load = np.bincount(share, weights=units, minlength=512)
        assert load.max() <= 1.15 * units.sum() / 512 + units.max(), (s, load.max(), units.sum() / 512, units.max())
    assert big >= 3  # W and U_b of the top node, U U^T
    kinds = {int(k) for k in plan[:, 1]}
    assert 0 in kinds and 2 in kinds  # 128 x 128 tiles where a launch fills the chip, 64 x 64 at the low levels of the tree
    last = plan[steps == int(steps.max())]
    assert (last[:, 6] >= 2).all() and last[:, 8].sum() // TILE == 5984 and len(last) >= 512  # U U^T: 528 tiles, 5984 units
    assert launches == 2 * 4 + 1 == plan[:, 0].max() + 1  # W and U_b of the four heights, then U U^T — and nothing else (round 5: 17)
