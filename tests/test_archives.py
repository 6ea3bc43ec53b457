"""limbo's on-disk formats ACROSS implementations ("checkpoint/resume compatibility with files written by stock
limbo"; VERDICT r4 missing 2).  Stock limbo here = /root/reference/src compiled unmodified into oracle/_ref/libref.so; its own
GP::save<TextArchive|BinaryArchive>(dir) / GP::load<...>(dir, recompute) (gp.hpp:439-511, serialize/text_archive.hpp:63-151,
binary_archive.hpp:66-161) write and read REAL files (boost::filesystem::create_directories from oracle/ref_build/shim).

  reference writes -> drop-in loads (recompute false and true) -> mu / sigma^2 vs the reference 1e-10 (test_serialize.cpp:159-177's bar)
  drop-in writes  -> reference loads (recompute false and true) -> the same
  binary files: what the drop-in writes after a recompute=false load is byte-identical to what the reference wrote (all six
  files); after compute() on the same data samples / observations / kernel_params / mean_params are byte-identical and
  matrixL / alpha agree to rounding.

Below Params::gpu::min_n_for_gpu() samples the drop-in's model lives on the host: those cases run without a GPU.  The -m gpu
case sends the same driver through the device (n = 300 > the default threshold of 256)."""
import filecmp
import os
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as OB

ROOT = Path(__file__).resolve().parent.parent
DRIVER = ROOT / "tests" / "cpp" / "test_archives"
NAMES = ("kernel_params", "mean_params", "samples", "observations", "matrixL", "alpha")

pytestmark = pytest.mark.skipif(not OB.ref_available(), reason="no oracle/_ref/libref.so")


def _driver(*args, env=None):
    if not DRIVER.exists():
        subprocess.check_call(["make", "-s", "-C", str(ROOT / "tests" / "cpp"), "test_archives"])
    r = subprocess.run([str(DRIVER), *[str(a) for a in args]], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    out = {}
    for ln in r.stdout.splitlines():
        w = ln.split()
        if w and w[0] in ("h_params", "log_lik", "mu", "sigma", "ucb", "gp_ucb", "ei"):
            out[w[0]] = np.array([float(v) for v in w[1:]])
        elif w and w[0] == "n":
            out["n"], out["dim_in"], out["dim_out"] = int(w[1]), int(w[3]), int(w[5])
    return out


def _read_bin_matrix(path):
    """binary_archive.hpp:144-150: Index rows, Index cols (8 bytes each), rows*cols doubles, column-major."""
    b = Path(path).read_bytes()
    r, c = struct.unpack("<qq", b[:16])
    assert len(b) == 16 + 8 * r * c
    return np.frombuffer(b[16:], dtype="<f8").reshape(r, c, order="F")


def _problem(kind, mean, n, D, P, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(n, D))
    Y = np.stack([np.cos((p + 1.5) * X.sum(axis=1)) + 0.3 * X[:, 0] for p in range(P)], axis=1) + 0.05 * rng.normal(size=(n, P))
    Q = np.concatenate([rng.uniform(-1, 1, size=(6, D)), X[:2]])
    theta = rng.uniform(-0.5, 0.5, size=D + 1 if kind == 0 else 2)
    return X, Y, Q, theta


def _ref(kind, mean, D, P):
    return OB.RefGP(kind, D, P, mean={0: OB.MEAN_DATA, 2: OB.MEAN_CONSTANT}[mean], noise=0.01, constant=1.0)


def _close(got, mu_r, s2_r, P, tol=1e-10):
    mu = got["mu"].reshape(-1, P)
    assert np.max(np.abs(mu - mu_r)) <= tol * max(1.0, np.max(np.abs(mu_r)))
    assert np.max(np.abs(got["sigma"] - s2_r) / s2_r) <= tol * 100  # sigma^2 ~ noise at the two training points: 1e-8 relative


def _cross(tmp_path, kind, mean, n, D, P, env=None):
    X, Y, Q, theta = _problem(kind, mean, n, D, P, seed=7 * kind + mean + n)
    qf = tmp_path / "q.txt"
    qf.write_text(f"{len(Q)}\n" + "\n".join(" ".join(repr(float(v)) for v in q) for q in Q) + "\n")
    # ---- stock limbo writes
    r = _ref(kind, mean, D, P)
    r.compute(X, Y)
    r.set_h_params(theta)
    if mean == 2:
        r.set_mean_h_params(np.linspace(0.3, 0.7, P))
    r.recompute(True, True)
    mu_r, s2_r = r.query(Q)
    ll_r = r.log_lik()
    for fmt in ("text", "bin"):
        r.save(tmp_path / "ref" / fmt, fmt == "bin")
    present = sorted(p.stem for p in (tmp_path / "ref" / "bin").iterdir())
    assert present == sorted(nm for nm in NAMES if nm != "mean_params" or mean == 2)
    # ---- the drop-in loads them, answers queries, and writes its own
    for fmt in ("text", "bin"):
        for rec in (0, 1):
            out_dir = tmp_path / f"dropin_{fmt}_{rec}"
            got = _driver("load", kind, mean, tmp_path / "ref" / fmt, fmt, rec, qf, out_dir, env=env)
            assert (got["n"], got["dim_in"], got["dim_out"]) == (n, D, P)
            assert np.max(np.abs(got["h_params"] - theta)) <= (0.0 if fmt == "bin" else 1e-14)
            _close(got, mu_r, s2_r, P)
            assert abs(got["log_lik"][0] - ll_r) <= 1e-10 * abs(ll_r)
            for nm in present:
                a, b = tmp_path / "ref" / "bin" / f"{nm}.bin", out_dir / "bin" / f"{nm}.bin"
                if fmt == "bin" and (rec == 0 or nm not in ("matrixL", "alpha")):
                    assert filecmp.cmp(a, b, shallow=False), (fmt, rec, nm)  # byte for byte what stock limbo wrote
                if nm in ("matrixL", "alpha"):
                    A, B = _read_bin_matrix(a), _read_bin_matrix(b)
                    assert A.shape == B.shape and np.max(np.abs(np.tril(A) - np.tril(B))) <= 1e-9 * np.max(np.abs(A)), (fmt, rec, nm)
            # ---- and stock limbo reads what the drop-in wrote, in both formats, with and without recompute
            for fmt2 in ("text", "bin"):
                for rec2 in (0, 1):
                    r2 = _ref(kind, mean, D, P)
                    r2.load(out_dir / fmt2, fmt2 == "bin", rec2)
                    assert r2.N == n and np.max(np.abs(r2.h_params() - theta)) <= 1e-14
                    mu2, s22 = r2.query(Q)
                    assert np.max(np.abs(mu2 - mu_r)) <= 1e-10 * max(1.0, np.max(np.abs(mu_r))), (fmt, rec, fmt2, rec2)
                    assert np.max(np.abs(s22 - s2_r) / s2_r) <= 1e-8, (fmt, rec, fmt2, rec2)
                    r2.close()
    r.close()
    # ---- a model the drop-in computed itself (default hyper-parameters), read by stock limbo
    df = tmp_path / "data.txt"
    with open(df, "w") as fh:
        fh.write(f"{P} {D} {n} {len(Q)}\n")
        for i in range(n):
            fh.write(" ".join(repr(float(v)) for v in list(X[i]) + list(Y[i])) + "\n")
        for q in Q:
            fh.write(" ".join(repr(float(v)) for v in q) + "\n")
    got = _driver("compute", kind, mean, df, tmp_path / "own", env=env)
    rc = _ref(kind, mean, D, P)
    rc.compute(X, Y)
    mu_c, s2_c = rc.query(Q)
    _close(got, mu_c, s2_c, P)
    rc.save(tmp_path / "refc", True)
    for nm in ("samples", "observations", "kernel_params"):
        assert filecmp.cmp(tmp_path / "refc" / f"{nm}.bin", tmp_path / "own" / "bin" / f"{nm}.bin", shallow=False), nm
    for fmt2 in ("text", "bin"):
        for rec2 in (0, 1):
            r2 = _ref(kind, mean, D, P)
            r2.load(tmp_path / "own" / fmt2, fmt2 == "bin", rec2)
            mu2, s22 = r2.query(Q)
            assert np.max(np.abs(mu2 - mu_c)) <= 1e-10 * max(1.0, np.max(np.abs(mu_c))), (fmt2, rec2)
            assert np.max(np.abs(s22 - s2_c) / s2_c) <= 1e-8, (fmt2, rec2)
            if not rec2:  # the factor stock limbo now holds is the drop-in's, bit for bit (binary) / to 15 digits (text)
                L_own = _read_bin_matrix(tmp_path / "own" / "bin" / "matrixL.bin")
                assert np.max(np.abs(r2.matrixL() - L_own)) <= (0.0 if fmt2 == "bin" else 1e-14 * np.max(np.abs(L_own)))
            r2.close()
    rc.close()


def _acqui_cross(tmp_path, kind, mean, n, D, P, M, env=None):
    """SURVEY 8f N1 ("the acquisition layer itself is only self-compared"): limbo's own acqui::UCB / GP_UCB / EI
    functors (acqui/ucb.hpp:83-90, gp_ucb.hpp:86-103, ei.hpp:85-116) over limbo's own model, point by point, against the
    drop-in's batch() over the SAME model — handed over through limbo's binary archive (recompute = false: the same factor
    and alpha bit for bit).  1e-10 on UCB / GP_UCB (of the value), 1e-9 absolute on EI (values ~1e-2 .. 1e-1)."""
    X, Y, _, theta = _problem(kind, mean, n, D, P, seed=900 + n)
    rng = np.random.default_rng(n)
    Q = np.concatenate([rng.uniform(-1, 1, size=(M - 3, D)), X[:3]])  # three AT training points (sigma ~ noise: EI's small-sigma branch)
    qf = tmp_path / "q.txt"
    qf.write_text(f"{len(Q)}\n" + "\n".join(" ".join(repr(float(v)) for v in q) for q in Q) + "\n")
    r = _ref(kind, mean, D, P)
    r.compute(X, Y)
    r.set_h_params(theta)
    r.recompute(True, True)
    r.save(tmp_path / "model", True)
    got = _driver("acqui", kind, mean, tmp_path / "model", "bin", qf, 7, env=env)
    for which, name, tol in ((0, "ucb", 1e-10), (1, "gp_ucb", 1e-10), (2, "ei", None)):
        want = r.acqui(which, Q, iteration=7)
        assert got[name].shape == want.shape
        if tol is not None:
            assert np.max(np.abs(got[name] - want) / np.maximum(np.abs(want), 1.0)) <= tol, name
        else:
            assert np.max(np.abs(got[name] - want)) <= 1e-9 and (got[name] >= 0).all() and got[name].max() > 0, name
    r.close()


def test_acquisition_batch_vs_limbos_functors_host_model(tmp_path):
    _acqui_cross(tmp_path, 0, 0, 60, 3, 1, M=40)        # 40 x 60 products: the drop-in's host loop
    _acqui_cross(tmp_path, 1, 0, 45, 2, 2, M=20)


@pytest.mark.gpu
def test_gpu_acquisition_batch_vs_limbos_functors(tmp_path):
    _acqui_cross(tmp_path, 0, 0, 300, 6, 1, M=500, env={"LIMBO_AMD_MIN_N_FOR_GPU": "0"})   # one device batch (query_batch)
    _acqui_cross(tmp_path, 0, 0, 200, 3, 1, M=300)                                          # host model, batch from its device shadow


@pytest.mark.parametrize("kind,mean,n,D,P", [(0, 0, 60, 3, 1), (1, 2, 45, 2, 2), (1, 0, 33, 4, 1)])
def test_archives_cross_implementation_host_model(tmp_path, kind, mean, n, D, P):
    """n below the drop-in's GPU threshold: the model is a host model, no device is touched (runs without a GPU)."""
    _cross(tmp_path, kind, mean, n, D, P)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mean,n,D,P", [(0, 0, 300, 6, 1), (1, 2, 130, 3, 2)])
def test_gpu_archives_cross_implementation(tmp_path, kind, mean, n, D, P):
    """The same exchange with the model on the device (LIMBO_AMD_MIN_N_FOR_GPU=0): load(recompute=false) uploads the stored
    factor (gpe_set_L / gpe_set_alpha), save() reads it back."""
    _cross(tmp_path, kind, mean, n, D, P, env={"LIMBO_AMD_MIN_N_FOR_GPU": "0"})
