"""Round-4 schedule A/B on one box: compute()+log_lik at N = 4096 (and other sizes) under different splits between the tall
data-flow launch, the one update behind it and the closing launch (GPE_TALL / GPE_TAIL_MAX are read per handle), the phase
times of the default schedule, and the batched forms (GPE_BATCH_TAIL is process-wide: child processes).

    python tools/r4_ab.py single | phases | sizes | batch [0|1] | all
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from limbo_amd import _capi, synth as O  # noqa: E402

# R4_SO: an A/B copy of the library (tools/build_variant.sh) instead of limbo_amd/libgpengine.so
eng = _capi.Lib(os.environ["R4_SO"], "gpe_") if os.environ.get("R4_SO") else _capi.load_engine()


def handle(X, om, kind, th, tall=None, tail=None):
    for k, v in (("GPE_TALL", tall), ("GPE_TAIL_MAX", tail)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    h = _capi.Handle(eng, 0)
    h.set_kernel(kind, th, 0.01)
    h.set_data(X, om)
    os.environ.pop("GPE_TALL", None)
    os.environ.pop("GPE_TAIL_MAX", None)
    return h


def timed(h, steps=30, warm=4):
    for _ in range(warm):
        h.compute()
        h.log_lik()
    per = []
    for _ in range(steps):
        t0 = time.perf_counter()
        info = h.compute()
        ll = h.log_lik()
        per.append(time.perf_counter() - t0)
    assert info == 0
    return 1e3 * float(np.median(per)), 1e3 * float(np.min(per)), ll


def single():
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    th = np.zeros(7)
    ref = None
    print("# N = 4096: median / min ms per compute()+log_lik; tall = columns of the tall launch (0: round-3 panels), tail = closing launch")
    for tall, tail in [(0, 2560), (4096, 2560), (4096, 2816), (4096, 3072), (4096, 2304), (4096, 2048), (4096, 1536), (0, 2560), (4096, 2560)]:
        h = handle(X, om, O.SE_ARD, th, tall, tail)
        med, mn, ll = timed(h)
        ref = ll if ref is None else ref
        print(f"tall_max {tall:5d} tail_max {tail:5d}: {med:.3f} / {mn:.3f} ms  -> {1e3 / med:7.1f} evaluations/s   log_lik rel diff {abs(ll - ref) / abs(ref):.1e}  reruns {h.flow_retries()}")
        h.close()


def single2():
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("GPE_TAIL_W", "GPE_TAIL_LAG", "GPE_TAIL_W_BATCH") if k in os.environ)
    for tall, tail in [(4096, 2560), (4096, 2816), (0, 2560)]:
        h = handle(X, om, O.SE_ARD, np.zeros(7), tall, tail)
        med, mn, ll = timed(h, steps=20, warm=3)
        print(f"[{tag}] tall_max {tall:5d} tail_max {tail:5d}: {med:.3f} / {mn:.3f} ms  -> {1e3 / med:7.1f} evaluations/s  log_lik {ll:.12g} reruns {h.flow_retries()}")
        h.close()
    for N in (2048, 1024):
        X2, Y2 = O.make_problem("c2", N=N)
        om2, _ = O.obs_mean_data(Y2)
        h = handle(X2, om2, O.SE_ARD, np.zeros(7), None, None)
        med, mn, ll = timed(h, steps=20, warm=3)
        print(f"[{tag}] N {N}: {med:.3f} / {mn:.3f} ms  log_lik {ll:.12g}")
        h.close()


def phases():
    X, Y = O.make_problem("c2", N=4096)
    om, _ = O.obs_mean_data(Y)
    for tall, tail in [(None, None), (4096, 2816), (0, 2560)]:
        h = handle(X, om, O.SE_ARD, np.zeros(7), tall, tail)
        h.compute()
        h.set_profiling(True)
        h.reset_phase_ms()
        reps = 5
        for _ in range(reps):
            h.compute()
            h.log_lik()
        ph = h.get_phase_ms()
        h.set_profiling(False)
        print(f"# phases (profiling mode: every phase alone between two events), tall {tall} tail {tail}")
        for k, v in ph.items():
            if v["launches"]:
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                print(f"  {k:14s} {1e3 * v['ms'] / reps:8.1f} us/step  launches/step {v['launches'] / reps:5.1f}  {v['flops'] / reps:.3e} flop  {tf:6.2f} TFLOP/s = {tf / 78.6:.3f}")
        h.close()


def sizes():
    print("# other sizes: ms per compute()+log_lik, tall launch on (default) / off (GPE_TALL=0)")
    for N in (2624, 3072, 3584, 5000, 6144, 8192):
        X, Y = O.make_problem("c2", N=N)
        om, _ = O.obs_mean_data(Y)
        row = []
        for tall in (None, 0):
            h = handle(X, om, O.SE_ARD, np.zeros(7), tall, None)
            med, mn, ll = timed(h, steps=12, warm=3)
            row.append((med, ll))
            h.close()
        print(f"N {N:5d}: default {row[0][0]:.3f} ms, GPE_TALL=0 {row[1][0]:.3f} ms, log_lik rel diff {abs(row[0][1] - row[1][1]) / abs(row[1][1]):.1e}")
    X3, Y3 = O.make_problem("c3")
    om3, _ = O.obs_mean_data(Y3)
    for tall in (None, 0):
        h = handle(X3, om3, O.MATERN52, np.zeros(2), tall, None)
        med, mn, ll = timed(h, steps=4, warm=2)
        print(f"N 16384 (c3) tall {tall}: {med:.2f} ms  log_lik {ll:.10g}")
        h.close()


def batch():
    print(f"# batched launches, GPE_BATCH_TAIL={os.environ.get('GPE_BATCH_TAIL', '1')} GPE_TAIL_MAX={os.environ.get('GPE_TAIL_MAX', '-')} GPE_TALL={os.environ.get('GPE_TALL', '-')}")
    for G, N in (((8, 2048),) if os.environ.get("R4_BATCH8") else ((8, 2048), (64, 2048), (10, 4096))):
        X, Y = O.make_problem("c4" if N == 2048 else "c2", N=N)
        rng = np.random.default_rng(4)
        hs = []
        for g in range(G):
            om, _ = O.obs_mean_data(Y * rng.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X[:, g % 6: g % 6 + 1] + g))
            h = _capi.Handle(eng, 0)
            h.set_data(X, om)
            h.set_kernel(O.SE_ARD, rng.uniform(-1e-2, 1e-2, size=7), 0.01)
            hs.append(h)
        _capi.batch_compute(hs)
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            st = _capi.batch_compute(hs)
            ll = _capi.batch_log_lik(hs)
        dt = (time.perf_counter() - t0) / reps
        fl = N ** 3 / 3.0 + 2.0 * N * N
        assert all(s == 0 for s in st)
        print(f"batch_compute   G {G:3d} N {N}: {1e3 * dt:8.3f} ms/batch  {G / dt:9.1f} evaluations/s  {G * fl / dt / 78.6e12:.3f} of peak  ll[0] {ll[0]:.12g} reruns {sum(h.flow_retries() for h in hs)}")
        if os.environ.get("R4_BATCH8"):  # the same members as G host threads, one launch chain each
            import threading
            def worker(h):
                for _ in range(reps):
                    h.compute()
                    h.log_lik()
            for h in hs:
                h.compute()
            ths = [threading.Thread(target=worker, args=(h,)) for h in hs]
            t0 = time.perf_counter()
            [t.start() for t in ths]
            [t.join() for t in ths]
            dtt = (time.perf_counter() - t0) / reps
            print(f"  the same as {G} host threads, one chain each: {1e3 * dtt:8.3f} ms/batch  reruns {[h.handover_reruns() for h in hs]} retries {[h.flow_retries() for h in hs]}")
        if (G, N) != (8, 2048):
            th = rng.uniform(-1e-2, 1e-2, size=(G, 7))
            _capi.batch_hp_objective(hs, O.SE_ARD, th, 0.01, want_grad=True)
            reps = 3
            t0 = time.perf_counter()
            for r in range(reps):
                lk, gr, st = _capi.batch_hp_objective(hs, O.SE_ARD, th + 1e-3 * (r + 1), 0.01, want_grad=True)
            dt = (time.perf_counter() - t0) / reps
            print(f"batch_hp_objective G {G:3d} N {N}: {1e3 * dt:8.3f} ms/batch  {G / dt:9.1f} objective evaluations/s  {G * float(N) ** 3 / dt / 78.6e12:.3f} of peak  lik[0] {lk[0]:.12g}")
        for h in hs:
            h.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("single", "all"):
        single()
    if what in ("phases", "all"):
        phases()
    if what in ("sizes", "all"):
        sizes()
    if what == "batch":
        batch()
    if what == "single2":
        single2()
    if what == "probe":  # one number: evaluations/s at N = 4096 under the defaults (tools/refresh_profiles.sh looks at it first)
        X, Y = O.make_problem("c2", N=4096)
        om, _ = O.obs_mean_data(Y)
        h = handle(X, om, O.SE_ARD, np.zeros(7), None, None)
        med, mn, ll = timed(h, steps=12, warm=3)
        print(f"{1e3 / med:.1f}")
    if what == "all":
        for env in ({"GPE_BATCH_TAIL": "0"}, {}, {"GPE_TAIL_MAX": "4096"}, {"GPE_TAIL_MAX": "2816"}):
            r = subprocess.run([sys.executable, __file__, "batch"], env=dict(os.environ, **env), capture_output=True, text=True)
            print(r.stdout + r.stderr[-2000:])
