"""The drop-in's host path (include/limbo_amd/limbo/model/gp/host_small.hpp): a model::GP below Params::gpu::min_n_for_gpu()
samples keeps its factor on the host and never touches the device — so these cases run without a GPU.  What the C++ driver
prints (tests/cpp/test_host_path) is held to the reference itself (limbo::model::GP compiled from /root/reference,
oracle/_ref) and to the C oracle: L, alpha, log-lik, mu, sigma^2 after compute() and after the add_sample() loop."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from limbo_amd import _capi, synth
from oracle import binding as OB
from oracle import np_oracle as O

ROOT = Path(__file__).resolve().parent.parent
DRIVER = ROOT / "tests" / "cpp" / "test_host_path"


def _run(tmp_path, kind, mean, X, Y, n0, Q):
    if not DRIVER.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "tests" / "cpp"), "test_host_path"])
    n1, D = X.shape
    P = Y.shape[1]
    f = tmp_path / "in.txt"
    with open(f, "w") as fh:
        fh.write(f"{kind} {mean} {P} {D} {n0} {n1} {len(Q)}\n")
        for i in range(n1):
            fh.write(" ".join(repr(float(v)) for v in list(X[i]) + list(Y[i])) + "\n")
        for q in Q:
            fh.write(" ".join(repr(float(v)) for v in q) + "\n")
    r = subprocess.run([str(DRIVER), str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out, cur = {}, None
    for ln in r.stdout.splitlines():
        w = ln.split()
        if w[0] in ("full", "incremental", "copy_recomputed"):
            cur = out.setdefault(w[0], {"n": int(w[2])})
        elif w[0] == "batch2":
            out["batch2"] = np.array([float(v) for v in w[1:]])
        else:
            cur[w[0]] = np.array([float(v) for v in w[1:]])
    return out


CASES = [(0, 0, 1, 3, 20, 45), (0, 1, 2, 2, 10, 70), (1, 0, 1, 4, 33, 64), (1, 2, 3, 2, 5, 40), (3, 0, 1, 2, 12, 30)]


@pytest.mark.skipif(not OB.ref_available(), reason="no oracle/_ref/libref.so")
@pytest.mark.parametrize("kind,mean,P,D,n0,n1", CASES)
def test_host_path_vs_reference_and_oracle(tmp_path, oracle_lib, kind, mean, P, D, n0, n1):
    rng = np.random.default_rng(100 * kind + 10 * mean + P)
    X = rng.uniform(-1, 1, size=(n1, D))
    Y = np.stack([np.cos((p + 1.5) * X.sum(axis=1)) + 0.3 * X[:, 0] for p in range(P)], axis=1) + 0.05 * rng.normal(size=(n1, P))
    Q = np.concatenate([rng.uniform(-1, 1, size=(4, D)), X[:2]])  # two of them AT training points (the clamp of gp.hpp:621-623)
    out = _run(tmp_path, kind, mean, X, Y, n0, Q)
    ref_mean = {0: OB.MEAN_DATA, 1: OB.MEAN_NULL, 2: OB.MEAN_CONSTANT}[mean]
    for tag, n in (("full", n0), ("incremental", n1), ("copy_recomputed", n1)):
        got = out[tag]
        assert got["n"] == n and got["status"][0] == 0
        r = OB.RefGP(kind, D, P, mean=ref_mean, noise=0.01, constant=1.0)
        if tag == "incremental":
            r.compute(X[:n0], Y[:n0])
            for i in range(n0, n1):
                r.add_sample(X[i], Y[i])
        else:
            r.compute(X[:n], Y[:n])
        Lr = r.matrixL()
        L = got["L"].reshape(n, n, order="F")
        assert np.max(np.abs(L - Lr)) <= 1e-12 * np.max(np.abs(Lr)), tag
        assert np.max(np.abs(got["alpha"].reshape(n, P, order="F") - r.alpha())) <= 1e-9 * np.max(np.abs(r.alpha())), tag
        assert abs(got["log_lik"][0] - r.log_lik()) <= 1e-11 * abs(r.log_lik()), tag
        mur, s2r = r.query(Q)
        assert np.max(np.abs(got["mu"].reshape(len(Q), P) - mur)) <= 1e-10 * max(1.0, np.max(np.abs(mur))), tag
        assert np.max(np.abs(got["sigma"] - s2r) / s2r) <= 1e-8, tag
        q = got["query"].reshape(len(Q), 2)
        assert np.array_equal(q[:, 0], got["mu"].reshape(len(Q), P)[:, 0]) and np.array_equal(q[:, 1], got["sigma"])
        r.close()
    # the C oracle on the final model (Data mean only: the oracle's obs_mean comes from the caller)
    if mean == 0:
        om, mvec = synth.obs_mean_data(Y)
        th = np.zeros(D + 1) if kind == 0 else np.zeros(2)
        o = _capi.Handle(oracle_lib)
        o.set_data(X, om)
        o.set_kernel(kind, th, 0.01)
        assert o.compute() == 0
        assert abs(out["copy_recomputed"]["log_lik"][0] - o.log_lik()) <= 1e-11 * abs(o.log_lik())
        o.close()
    b = out["batch2"]
    assert np.array_equal(b[:2], out["incremental"]["mu"].reshape(len(Q), P)[:2, 0]) and np.array_equal(b[2:], out["incremental"]["sigma"][:2])


def test_host_path_under_asan_ubsan(tmp_path):
    """SURVEY §5 (the reference's memory / race checks are host-side tooling): the drop-in header set with
    -fsanitize=address,undefined (`make -C tests/cpp asan`).  The host path needs no GPU, so the instrumented driver runs
    here: compute, the add_sample loop, copies and queries must finish without a sanitizer report."""
    import os

    subprocess.check_call(["make", "-s", "-C", str(ROOT / "tests" / "cpp"), "test_host_path_asan"])
    exe = ROOT / "tests" / "cpp" / "test_host_path_asan"
    rng = np.random.default_rng(1)
    n1, D, P, n0 = 48, 3, 2, 15
    X, Y, Q = rng.uniform(-1, 1, (n1, D)), rng.normal(size=(n1, P)), rng.uniform(-1, 1, (4, D))
    f = tmp_path / "in.txt"
    with open(f, "w") as fh:
        fh.write(f"0 0 {P} {D} {n0} {n1} {len(Q)}\n")
        for i in range(n1):
            fh.write(" ".join(repr(float(v)) for v in list(X[i]) + list(Y[i])) + "\n")
        for q in Q:
            fh.write(" ".join(repr(float(v)) for v in q) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe), str(f)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "incremental n 48" in r.stdout
