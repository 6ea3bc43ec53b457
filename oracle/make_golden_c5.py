"""Mint tests/golden/c5_truth_n200_noise1e-10.npz: 50-digit ground truth for BASELINE configs[4] at the benchmark's own
noise (src/benchmarks/limbo/bench.cpp:70: noise 1e-10; :66-67,:83-84: 10 random samples + 190 add_sample calls, Hartmann6).

Why: at noise 1e-10 cond(K) ~ sigma_f^2 n / (noise + 1e-8) ~ 2e10, and two correct fp64 implementations that sum in a
different order differ by ~cond * eps.  tests/test_gpu_configs.py::test_gpu_c5_add_sample_loop therefore held the engine
to the reference only at L 1e-6 / mu 1e-4 / sigma^2 1e-6 — which does not show that the engine is
no farther from the TRUTH than the reference is.  This fixture is the truth: K in 50 digits from the double inputs
(gp.hpp:550-562, kernel.hpp:81-84, squared_exp_ard.hpp:138-151), L = chol(K) (gp.hpp:565), alpha (gp.hpp:605-611),
mu / sigma^2 on 32 points (gp.hpp:613-624, :166), each rounded once to double at the end.  The GPU test then asserts
|gpu - truth| <= 4 |reference - truth| (max-norm) for L, mu and sigma^2.

Same draws as the test (numpy default_rng(2026): X 200 x 6, then Xq 32 x 6).  Run: python oracle/make_golden_c5.py (~2 min).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from limbo_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden" / "c5_truth_n200_noise1e-10.npz"


def main():
    import mpmath as mp

    mp.mp.dps = 60
    rng = np.random.default_rng(2026)
    n, D, noise = 200, 6, 1e-10
    X = rng.uniform(0, 1, size=(n, D))
    Y = synth.hartmann6(X)[:, None]
    Xq = rng.uniform(0, 1, size=(32, D))
    th = np.zeros(D + 1)
    om, mean = synth.obs_mean_data(Y)
    Xm = [[mp.mpf(float(v)) for v in row] for row in X]
    Qm = [[mp.mpf(float(v)) for v in row] for row in Xq]
    ell = [mp.exp(mp.mpf(float(t))) for t in th[:D]]
    sf2 = mp.exp(2 * mp.mpf(float(th[D])))

    def kf(a, b):
        z = mp.mpf(0)
        for d in range(D):
            q = (a[d] - b[d]) / ell[d]
            z += q * q
        return sf2 * mp.exp(-z / 2)

    jit = mp.mpf(float(noise)) + mp.mpf(float(1e-8))  # the doubles the reference adds (kernel.hpp:83)
    # right-looking Cholesky on lists of mpf (mp.cholesky on an mp.matrix is ~3x slower)
    A = [[kf(Xm[i], Xm[j]) + (jit if i == j else 0) for j in range(i + 1)] for i in range(n)]
    L = [[mp.mpf(0)] * n for _ in range(n)]
    for j in range(n):
        s = A[j][j] - mp.fsum(L[j][k] * L[j][k] for k in range(j))
        L[j][j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i][j] = (A[i][j] - mp.fsum(L[i][k] * L[j][k] for k in range(j))) / L[j][j]
    b = [mp.mpf(float(v)) for v in om[:, 0]]

    def fwd(r):
        y = [mp.mpf(0)] * n
        for i in range(n):
            y[i] = (r[i] - mp.fsum(L[i][k] * y[k] for k in range(i))) / L[i][i]
        return y

    def bwd(r):
        a = [mp.mpf(0)] * n
        for i in range(n - 1, -1, -1):
            a[i] = (r[i] - mp.fsum(L[k][i] * a[k] for k in range(i + 1, n))) / L[i][i]
        return a

    alpha = bwd(fwd(b))
    kta, var = [], []
    for q in Qm:
        ks = [kf(Xm[i], q) for i in range(n)]
        z = fwd(ks)
        kta.append(mp.fsum(ks[i] * alpha[i] for i in range(n)))
        var.append(kf(q, q) - mp.fsum(v * v for v in z))
    kta = np.array([[float(v)] for v in kta])
    var = np.array([float(v) for v in var])
    mu, s2 = synth.finish_query(kta, var, mean, noise)
    Ld = np.array([[float(L[i][j]) for j in range(n)] for i in range(n)])
    ll = -mp.fsum(b[i] * alpha[i] for i in range(n)) / 2 - mp.fsum(mp.log(L[i][i]) for i in range(n)) - mp.mpf(n) / 2 * mp.log(2 * mp.pi)
    np.savez_compressed(OUT, X=X, Y=Y, Xq=Xq, theta=th, noise=noise, L=Ld, alpha=np.array([float(v) for v in alpha]),
                        kta=kta, var_raw=var, mu=mu, sigma2=s2, log_lik=float(ll), source="mpmath-60")
    print("wrote", OUT.name, "log_lik", float(ll), "min sigma^2", s2.min())


if __name__ == "__main__":
    main()
