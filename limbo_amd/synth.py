"""Synthetic regression problems of BASELINE.json's configs and the host-side finish of a
query — shared by bench.py, the tools and the tests.  Pure numpy; no oracle, no device code.

The reference's own generators are entropy-seeded (tools/random_generator.hpp:83); seeds here are ours.
"""
import numpy as np

SE_ARD, MATERN52, MATERN32, EXP, HOST_K = range(5)  # include/gpe.h enum gpe_kernel_kind


def finish_query(kta, var_raw, mean_at_v, noise):
    """What the C++ wrapper applies on the host: mu = k^T alpha + m(v) (gp.hpp:615);
    sigma^2 = (res <= eps ? 0 : res) + noise (gp.hpp:623, :166)."""
    eps = np.finfo(float).eps
    var = np.where(var_raw <= eps, 0.0, var_raw) + noise
    return kta + mean_at_v, var



def hartmann6(X):
    """src/benchmarks/regression/test_functions.hpp:343-366."""
    a = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8],
                  [17, 8, 0.05, 10, 0.1, 14]])
    p = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                  [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.665], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
    al = np.array([1.0, 1.2, 3.0, 3.2])
    X = np.atleast_2d(X)
    s = np.einsum("ij,nij->ni", a, (X[:, None, :] - p[None]) ** 2)
    return (al * np.exp(-s)).sum(axis=1)


def rastrigin(X):
    """src/benchmarks/regression/test_functions.hpp:50-66 (A = 10)."""
    X = np.atleast_2d(X)
    return 10.0 * X.shape[1] + np.sum(X * X - 10.0 * np.cos(2 * np.pi * X), axis=1)


def make_problem(config, seed=None, N=None, D=None, P=1):
    """Synthetic (X, y) per SURVEY.md §8(d).  numpy default_rng(20260925 + config index)
    replaces the reference's entropy-seeded RNG (tools/random_generator.hpp:83).
    Noise rule: y += N(0, (std(y)/20)^2) (waf_tools/benchmark_template.cpp:112-120)."""
    idx = {"c1": 1, "c2": 2, "c3": 3, "c4": 4, "c5": 5}[config]
    rng = np.random.default_rng(20260925 + idx if seed is None else seed)
    if config == "c1":
        N = N or 200
        D = D or 2
        X = rng.uniform(-5.12, 5.12, size=(N, D))
        y = rastrigin(X)
    elif config in ("c2", "c4", "c5"):
        N = N or (4096 if config == "c2" else 2048)
        D = 6
        X = rng.uniform(0.0, 1.0, size=(N, D))
        y = hartmann6(X)
    else:
        N = N or 16384
        D = D or 12
        X = rng.uniform(0.0, 1.0, size=(N, D))
        y = np.cos(2 * X).sum(axis=1)
    y = y + rng.normal(0.0, np.std(y, ddof=1) / 20.0, size=N)
    Y = y[:, None]
    if P > 1:
        Y = np.concatenate([Y] + [(y * rng.uniform(0.5, 1.5) + rng.normal(0, 0.05, N))[:, None] for _ in range(P - 1)], axis=1)
    return X, Y


def obs_mean_data(Y):
    """mean::Data (mean/data.hpp:59-63) + gp.hpp:111,:537-548: obs_mean = Y - colwise mean."""
    Y = np.asarray(Y, float).reshape(len(Y), -1)
    m = Y.mean(axis=0)
    return Y - m, m
