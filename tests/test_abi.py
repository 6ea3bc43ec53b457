"""CPU tests of the boundary: libgpengine.so loads and exports every symbol include/gpe.h
declares; the oracle mirrors the same set under orc_.  No compute calls (no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "gpe.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpe_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("gpe_compute", "gpe_add_sample", "gpe_query_batch", "gpe_log_lik", "gpe_log_lik_grad",
                 "gpe_hp_objective", "gpe_clone", "gpe_set_data"):
        assert must in syms


def test_engine_exports_every_declared_symbol():
    from limbo_amd import _capi

    assert _capi.ENGINE_SO.exists(), "libgpengine.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.gpe_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.gpe_version()


def test_engine_contains_gfx950_code_object():
    from limbo_amd import _capi

    blob = _capi.ENGINE_SO.read_bytes()
    assert b"gfx950" in blob
    assert b"k_gemm4" in blob and b"k_gemm_glds" in blob and b"k_diag" in blob
    # every hand-written kernel family of the path is in the shipped code object
    for k in (b"k_build", b"k_panel_step", b"k_upd_fused", b"k_trsv_bwd_flow", b"k_trsv_fwd_flow", b"k_trsv_bwd_flow_mp",
              b"k_trsv_fwd_flow_mp", b"k_inv_panels", b"k_grad_tiles", b"k_lambda_rows", b"k_loo_prep"):
        assert k in blob, k


def test_engine_is_independent_of_the_oracle():
    """The product must not link or reference the oracle."""
    from limbo_amd import _capi

    blob = _capi.ENGINE_SO.read_bytes()
    assert b"liboracle" not in blob and b"orc_compute" not in blob
    # no product source — device code, C-ABI header, drop-in headers, Python package — imports, includes, links, dlopens or
    # spawns anything under oracle/ (comments may name the oracle: they are stripped first)
    import re

    offenders = []
    files = [p for root in (ROOT / "limbo_amd", ROOT / "include") for p in root.rglob("*")
             if p.is_file() and (p.suffix in (".hip", ".h", ".hpp", ".py", ".cpp", ".c") or p.name == "Makefile")]
    assert len(files) > 30
    for p in files:
        text = p.read_text()
        if p.suffix == ".py":
            text = re.sub(r'"""[\s\S]*?"""', "", text)
            text = re.sub(r"#[^\n]*", "", text)
        else:
            text = re.sub(r"/\*[\s\S]*?\*/", "", text)
            text = re.sub(r"//[^\n]*", "", text)
            text = re.sub(r"^\s*#(?!include)[^\n]*", "", text, flags=re.M) if p.name == "Makefile" else text
        for pat in (r"\borc_\w+", r"liboracle", r"gp_oracle", r"np_oracle", r"from\s+oracle\b", r"import\s+oracle\b",
                    r"oracle/", r"libref\.so", r"_ref/"):
            if re.search(pat, text):
                offenders.append((str(p.relative_to(ROOT)), pat))
    assert not offenders, offenders


def test_oracle_mirrors_the_abi(oracle_lib):
    skip = {"gpe_get_stream", "gpe_set_profiling", "gpe_get_phase_ms", "gpe_reset_phase_ms", "gpe_mfma_f64_peak",
            "gpe_hbm_stream_peak", "gpe_flow_retries", "gpe_small_calls", "gpe_handover_reruns", "gpe_trace", "gpe_trace_dump",
            "gpe_debug_tail_order", "gpe_debug_tail_plan", "gpe_debug_inv_plan", "gpe_debug_chain_split", "gpe_debug_ragged_split",
            "gpe_debug_tri_tile_map", "gpe_epoch", "gpe_xproc_waits"}  # (engine bookkeeping, nothing of the reference's to restate)
    for s in declared_symbols():
        if s in skip:
            continue
        assert hasattr(oracle_lib.cdll, "orc_" + s[4:]), s


def test_binding_loads_engine_without_gpu():
    from limbo_amd import _capi

    lib = _capi.load_engine()  # declares argtypes for every symbol: AttributeError if one is missing
    assert lib.prefix == "gpe_"


def test_dispatch_tables_of_the_data_flow_launches_are_deadlock_free():
    """csrc/potrf.hip: tail_order.  A data-flow launch (k_tail) is deadlock-free because every workgroup waits for
    lower-numbered ones only; the order is a TABLE (diagonal workgroups early, tiles below the triangle lagging the chain).
    Built and checked on the host for every launch shape the engine can produce up to 64 tile columns — closing launches
    (nb = nt, nt + 1 with the right-hand-side strip), tall launches (nb up to 65), lags 0..5: a permutation of the tiles and, in the
    strict table (batched launches), no wait for a higher-numbered workgroup.  Round 6: single launches dispatch the chain
    workgroups four columns early (TAIL_DLEAD) — those may wait for a tile dispatched after them; the table is accepted only if
    at every point of the order at most 16 of them can be in that state (everything else in front of that point waits for
    lower-numbered workgroups only and makes progress on the other CUs).  The hook checks both tables."""
    from limbo_amd import _capi

    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_tail_order
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 4
    assert f(0, 1, 3, 0) == -1 and f(4, 3, 3, 0) == -1
    bad = []
    for nt in range(1, 65):
        for nb in sorted({nt, nt + 1, min(nt + 7, 65), 64, 65}):
            if nb < nt:
                continue
            for lag in (0, 2, 3, 5):
                if f(nt, nb, lag, 0) != 1:
                    bad.append((nt, nb, lag))
    assert not bad, bad[:10]
    assert f(8, 9, 3, 1) == -1  # (the two-blocks-per-chain-workgroup form of round 4 is gone)


def test_chain_workgroup_splits_its_products_evenly_and_completely():
    """csrc/potrf.hip: syrk40 / tri_solve32 (round 5).  The chain workgroup of a data-flow launch multiplies only what is kept: of
    the 64 units of 16 x 4 of a diagonal block's update Y Y^T, the 40 that touch the LOWER triangle — five per wave, every unit once —
    and, against the triangular 32 x 32 block inverses, k ranges that stop at a wave's last column, dealt so that the two waves of a
    SIMD (w and w + 4) add up to the same.  The device code's own mapping functions, called on the host."""
    from limbo_amd import _capi

    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_chain_split
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    units, cols = {}, []
    for w in range(8):
        u10, c = (ctypes.c_int * 10)(), ctypes.c_int()
        assert f(w, u10, ctypes.byref(c)) == 0
        mine = [(u10[2 * q], u10[2 * q + 1]) for q in range(5)]
        assert len(set(mine)) == 5
        for ij in mine:
            assert ij not in units, "a unit computed twice"
            units[ij] = w
        cols.append(c.value)
    assert f(8, (ctypes.c_int * 10)(), ctypes.byref(ctypes.c_int())) == -1
    # exactly the units (row block i of 16, column block j of 4) with an element on or below the diagonal: 16 i + 15 >= 4 j
    want = {(i, j) for i in range(4) for j in range(16) if 16 * i + 15 >= 4 * j}
    assert set(units) == want and len(want) == 40
    # the triangular products: the four column blocks of eight are all there, each for both row halves (waves 2 q, 2 q + 1), and the
    # k loops (cols + 8 steps of one) of the two waves of a SIMD add up to 40
    assert sorted(cols) == [0, 0, 8, 8, 16, 16, 24, 24] and all(cols[2 * q] == cols[2 * q + 1] for q in range(4))
    assert all((cols[w] + 8) + (cols[w + 4] + 8) == 40 for w in range(4))


def test_ragged_block_update_deals_its_k_range_completely():
    """csrc/potrf.hip: launch_ragged_update (round 5).  The last block of a ragged order is updated over everything the data-flow
    launch factored (k = 256 .. 2816) by up to 32 workgroups of kc rows each + an ordered fold: the chunks cover [0, k) exactly,
    the last one is not empty, kc is a multiple of the kernel's 32-row blocks, the slots fit the scratch, and small k / small
    scratch fall back to the general product."""
    from limbo_amd import _capi

    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_ragged_split
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
    kc = ctypes.c_int()
    assert f(192, 1 << 20, ctypes.byref(kc)) == 0 and f(2048, 4096, ctypes.byref(kc)) == 0 and f(2048, 1 << 20, None) == -1
    for k in range(256, 2816 + 1, 64):
        for scratch in (2 * 4096, 5 * 4096 + 17, 14 * 4096 + 4 * 3072, 1 << 22):
            G = f(k, scratch, ctypes.byref(kc))
            assert 1 <= G <= 32 and G * 4096 <= scratch, (k, scratch, G)
            assert kc.value % 32 == 0 and G * kc.value >= k and (G - 1) * kc.value < k, (k, scratch, G, kc.value)
    G = f(1664, 1 << 22, ctypes.byref(kc))  # N = 1700: 26 workgroups of 64 rows
    assert (G, kc.value) == (26, 64)


def test_triangular_update_tile_order_covers_every_tile_once_and_keeps_panels_few():
    """csrc/gemm.hip: tri_tile_map (round 6).  The 128 x 128 tiles of a triangular trailing update are dealt to the workgroups
    through a host-built table: every live tile (one that touches the lower triangle) exactly once, no other; the table is a whole
    number of rounds of the eight XCDs (workgroup b runs on XCD b mod 8); the tiles of an XCD are a run of the band order — for
    the update of N = 4096 (2816 + 1 rows behind column 1280, k = 1280) the eight XCDs fetch at most 96 distinct 128-row panels
    between them against 139 of the folded column order it replaces; shapes: square, ragged, with right-hand-side rows as a
    tile row, a diagonal that does not start at a tile corner (a ragged order's last block), more than one round."""
    from limbo_amd import _capi

    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_tri_tile_map
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    buf = (ctypes.c_int * 65536)()
    assert f(0, 128, 0, 0, buf, 65536) == -1 and f(128, 128, 0, 0, None, 0) == -1
    for m, n, r0, c0 in [(2817, 2816, 1280, 1280), (2816, 2816, 0, 0), (128, 128, 0, 0), (130, 100, 64, 64), (3000, 2900, 1024, 1024),
                         (2945, 2880, 1536, 1600), (14849, 14848, 1536, 1536), (700, 300, 4096, 4096), (257, 256, 64, 0)]:
        ln = f(m, n, r0, c0, buf, 65536)
        tm, tn = -(-m // 128), -(-n // 128)
        live = {(ti, tj) for tj in range(tn) for ti in range(tm) if r0 + ti * 128 + 127 >= c0 + tj * 128}
        assert ln % 8 == 0 and 0 < ln <= 65536 and ln < len(live) + 8, (m, n, ln)
        got = [(buf[b] & 0xffff, buf[b] >> 16) for b in range(ln) if buf[b] >= 0]
        assert len(got) == len(live) and set(got) == live, (m, n)
        per_xcd = [[(buf[b] & 0xffff, buf[b] >> 16) for b in range(x, ln, 8) if buf[b] >= 0] for x in range(8)]
        assert max(map(len, per_xcd)) - min(map(len, per_xcd)) <= 1
        if (m, n) == (2817, 2816):
            panels = sum(len({i for i, _ in t} | {j for _, j in t}) for t in per_xcd)
            assert len(live) == 253 + 22 and panels <= 110, panels  # (with the right-hand-side row as a tile row: 23 tile rows)
    # the same update with its right-hand-side row as FMAs in front of the tiles (what the engine launches): 253 tiles
    ln = f(2816, 2816, 1280, 1280, buf, 65536)
    per_xcd = [[(buf[b] & 0xffff, buf[b] >> 16) for b in range(x, ln, 8) if buf[b] >= 0] for x in range(8)]
    assert ln == 256 and sum(map(len, per_xcd)) == 253
    assert sum(len({i for i, _ in t} | {j for _, j in t}) for t in per_xcd) <= 96


def test_schedule_of_the_factorisation_by_size():
    """engine.hip: tail_plan (host logic, no device).  N <= 3328 (round 6; 2816 before): ONE data-flow launch from column 0; up to tall_max + tail_max =
    4352: tall launch from column 0 | one update | closing launch; larger: 256-column panels in front of the closing launch;
    the panels end on a panel boundary, the closing launch is never wider than tail_max, a ragged order rides as one more row
    strip.  Batched sequences: data-flow launches from column 0 or not at all, and only while the members' tiles fit."""
    from limbo_amd import _capi

    lib = ctypes.CDLL(str(_capi.ENGINE_SO))
    f = lib.gpe_debug_tail_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]

    def plan(n, p=1, g=1, tail=0, tall=0, btail=0):
        out = (ctypes.c_int64 * 8)()
        assert f(n, p, g, tail, tall, btail, out) == 0
        return dict(zip(("t0", "e0", "nt_tail", "nb_tail", "nt_tall", "nb_tall", "n64", "nbo"), list(out)))

    assert plan(4096) == dict(t0=1280, e0=0, nt_tail=44, nb_tail=45, nt_tall=20, nb_tall=65, n64=4096, nbo=256)
    assert plan(2048)["t0"] == 0 and plan(2048)["e0"] == -1 and plan(2048)["nt_tail"] == 32
    assert plan(2816)["t0"] == 0 and plan(2880)["t0"] == 0 and plan(3328)["t0"] == 0 and plan(3328)["nt_tail"] == 52
    assert plan(3392)["t0"] == 768 and plan(3392)["e0"] == 0  # (above the one-launch range: closing launch <= 2816 columns again)
    assert plan(2880, tail=2816)["t0"] == 256 and plan(2880, tail=2816)["e0"] == 0  # (an explicit width is the width used)
    assert plan(5000)["t0"] == 2304 and plan(5000)["e0"] == -1  # (the tall launch only from column 0 and <= 1536 wide)
    assert plan(100)["t0"] == -1 and plan(128)["t0"] == 0
    assert plan(4096, tail=2560)["t0"] == 1536 and plan(4096, tail=2560, tall=1024)["e0"] == -1
    assert plan(2048, g=8) == dict(t0=512, e0=0, nt_tail=24, nb_tail=25, nt_tall=8, nb_tall=33, n64=2048, nbo=256)
    assert plan(2048, g=64)["t0"] == -1 and plan(4096, g=10)["t0"] == -1  # (too many tiles for the chip: panels)
    assert plan(1024, g=16)["t0"] == 0
    for n in list(range(64, 9000, 61)) + [16384]:
        for p in (1, 3):
            pl = plan(n, p)
            if pl["t0"] < 0:  # fewer than two tile columns, or the ragged rows + obs_mean's rows do not fit ONE extra 64-row strip
                assert n < 128 or n - n // 64 * 64 + p > 64, (n, p)
                continue
            assert n - n // 64 * 64 + p <= 64
            assert pl["n64"] == n // 64 * 64 and pl["t0"] % pl["nbo"] == 0
            assert 2 <= pl["nt_tail"] == (pl["n64"] - pl["t0"]) // 64 <= (3328 if pl["t0"] == 0 else 2816) // 64
            assert pl["nb_tail"] == pl["nt_tail"] + 1  # (obs_mean's rows, and a ragged last block, as one more row strip)
            if pl["e0"] >= 0:
                assert pl["e0"] == 0 and 2 <= pl["nt_tall"] == pl["t0"] // 64 <= 1536 // 64 and pl["nb_tall"] == pl["n64"] // 64 + 1
            else:
                assert pl["nt_tall"] == 0 and (pl["t0"] == 0 or pl["t0"] > 1536)
