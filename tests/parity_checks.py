"""Parity bodies shared by test_oracle.py (CPU: oracle vs goldens) and test_gpu_parity.py
(GPU: HIP engine vs goldens and vs the oracle).  Tolerances are SURVEY.md §8(c)'s acceptance
bars: mu / sigma^2 <= 1e-8 rel (sigma^2 including +noise), log-lik <= 1e-10 rel,
grad <= 1e-6 rel; L and alpha are checked in norm.
"""
import numpy as np

from oracle import np_oracle as O
from tests.util import load, new_gp, relerr, relerr_norm

TOL_MU = 1e-8
TOL_VAR = 1e-8
TOL_LL = 1e-10
TOL_GRAD = 1e-6


def check_against_mp_golden(lib, path, tol_scale=1.0):
    g = load(path)
    h = new_gp(lib, g["kind"], g["X"], g["obs_mean"], g["theta"], g["noise"])
    assert h.compute() == 0
    N = g["X"].shape[0]
    L = h.get_L()
    assert np.all(np.triu(L, 1) == 0.0), "matrixL() must have a zero upper triangle (gp.hpp:411)"
    assert relerr_norm(L, g["L"]) < 1e-11 * tol_scale
    # alpha: forward error is bounded by cond(K) * eps
    assert relerr_norm(h.get_alpha(), g["alpha"]) < 1e-9 * tol_scale
    ll = h.log_lik()
    assert abs(ll - g["log_lik"]) <= TOL_LL * max(1.0, abs(g["log_lik"])) * tol_scale
    grad = h.log_lik_grad(bool(g["optimize_noise"]))
    assert relerr_norm(grad, g["grad"]) < TOL_GRAD * tol_scale
    kta, var = h.query_batch(g["Xq"])
    mu, s2 = O.finish_query(kta, var, g["mean"], g["noise"])
    mu_ref, s2_ref = O.finish_query(g["kta"], g["var_raw"], g["mean"], g["noise"])
    assert relerr(mu, mu_ref, floor=1e-3) < TOL_MU * tol_scale
    assert relerr(s2, s2_ref) < TOL_VAR * tol_scale
    Kinv = h.get_Kinv()
    assert relerr_norm(Kinv, g["Kinv"]) < 1e-9 * tol_scale
    assert N == h.nb_samples()
    h.close()


def check_against_np_golden(lib, path):
    g = load(path)
    X, Y = O.make_problem(g["config"], N=int(g["N"]))
    om, mean = O.obs_mean_data(Y)
    assert np.allclose(mean, g["mean"], rtol=0, atol=0), "synthetic problem generator drifted"
    h = new_gp(lib, g["kind"], X, om, g["theta"], g["noise"])
    assert h.compute() == 0
    L = h.get_L()
    idx = g["L_idx"]
    # absolute bar scaled by max|L|: tiny entries of L carry cond(K)*eps absolute error
    assert np.max(np.abs(L[idx[:, 0], idx[:, 1]] - g["L_samples"])) < 1e-10 * np.max(np.abs(g["L_diag"]))
    assert relerr(np.diag(L), g["L_diag"]) < 1e-10
    assert relerr_norm(h.get_alpha(), g["alpha"]) < 1e-7
    ll = h.log_lik()
    assert abs(ll - g["log_lik"]) <= TOL_LL * abs(g["log_lik"])
    grad = h.log_lik_grad(True)
    assert relerr_norm(grad, g["grad"]) < TOL_GRAD
    kta, var = h.query_batch(g["Xq"])
    mu, s2 = O.finish_query(kta, var, g["mean"], g["noise"])
    mu_ref, s2_ref = O.finish_query(g["kta"], g["var_raw"], g["mean"], g["noise"])
    assert relerr(mu, mu_ref, floor=1e-3) < TOL_MU
    assert relerr(s2, s2_ref) < TOL_VAR
    h.close()


def check_incremental_vs_full(lib, kind=O.SE_ARD, n0=5, n1=60, D=3, P=2, seed=3, dup=False):
    """test_gp.cpp:513-566 / :568-635: add_sample() one by one == compute() on all samples
    (matrixL isApprox 1e-5, mu within 1e-5); with `dup` the 10x duplicated-point
    near-singular K of test_gp.cpp:513-566."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(n1, D))
    if dup:
        X[n0:n0 + 10] = X[0]
    Y = np.stack([np.sin(X.sum(axis=1) * (p + 1)) for p in range(P)], axis=1)
    nt = D + 1 if kind == O.SE_ARD else 2
    theta = np.zeros(nt)
    noise = 0.01
    om0, _ = O.obs_mean_data(Y[:n0])
    h = new_gp(lib, kind, X[:n0], om0, theta, noise)
    assert h.compute() == 0
    for n in range(n0, n1):
        om, _ = O.obs_mean_data(Y[:n + 1])
        h.add_sample(X[n], om)
    assert h.nb_samples() == n1
    om, mean = O.obs_mean_data(Y)
    f = new_gp(lib, kind, X, om, theta, noise)
    assert f.compute() == 0
    Li, Lf = h.get_L(), f.get_L()
    assert np.allclose(Li, Lf, rtol=1e-5, atol=1e-8)          # isApprox(…, 1e-5)
    assert np.allclose(Li @ Li.T, f.get_K(), rtol=1e-5, atol=1e-8)  # L L^T ~ K (test_gp.cpp:553-561)
    Xq = rng.uniform(-1, 1, size=(20, D))
    ki, vi = h.query_batch(Xq)
    kf, vf = f.query_batch(Xq)
    assert np.max(np.abs(ki - kf)) < 1e-5
    assert np.max(np.abs(vi - vf)) < 1e-5
    assert abs(h.log_lik() - f.log_lik()) < 1e-6 * max(1.0, abs(f.log_lik()))
    h.close()
    f.close()


def check_add_sample_from_empty(lib):
    """gp.hpp:126-137: add_sample on an empty GP sets the dimensions."""
    from limbo_amd import _capi

    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, size=(6, 2))
    Y = rng.normal(size=(6, 1))
    h = _capi.Handle(lib)
    h.set_kernel(O.SE_ARD, np.zeros(3), 0.01)
    for n in range(6):
        om, _ = O.obs_mean_data(Y[:n + 1])
        h.add_sample(X[n], om)
    om, _ = O.obs_mean_data(Y)
    f = new_gp(lib, O.SE_ARD, X, om, np.zeros(3), 0.01)
    f.compute()
    assert np.allclose(h.get_L(), f.get_L(), rtol=1e-10, atol=1e-12)
    assert np.allclose(h.get_alpha(), f.get_alpha(), rtol=1e-8, atol=1e-10)
    h.close()
    f.close()


def check_grad_fd(lib, kind, optimize_noise, N=40, D=4, P=2, trials=8, seed=5, lam=0):
    """test_gp.cpp:131-271: analytic grad of the log-lik vs central finite differences of the
    log-lik THROUGH the HP objective (kernel_lf_opt.hpp:77-92), e = 1e-4."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(N, D))
    Y = np.stack([np.cos(X.sum(axis=1) * (p + 1)) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Y)
    nt = D + D * lam + 1 if kind == O.SE_ARD else 2  # lam columns of Lambda (squared_exp_ard.hpp:94)
    h = new_gp(lib, kind, X, om, np.zeros(nt), 0.01)
    tot = 0.0
    e = 1e-4
    for _ in range(trials):
        th = rng.uniform(-1.5, 1.5, size=nt + (1 if optimize_noise else 0))

        def f(t, want_grad=False):
            noise = np.exp(2 * t[nt]) if optimize_noise else 0.01
            return h.hp_objective(kind, t[:nt], noise, optimize_noise, want_grad)

        lik, g, _ = f(th, True)
        fd = np.zeros_like(th)
        for j in range(th.size):
            tp, tm = th.copy(), th.copy()
            tp[j] += e
            tm[j] -= e
            fd[j] = (f(tp)[0] - f(tm)[0]) / (2 * e)
        tot += np.linalg.norm(fd - g) / max(1.0, np.linalg.norm(g))
    assert tot < trials * 1e-4
    h.close()


def check_loo_grad_fd(lib, kind, optimize_noise, N=40, D=4, P=2, trials=6, seed=11):
    """test_gp.cpp:273-380: analytic gradient of the LOO-CV log probability vs central finite
    differences of compute_log_loo_cv after recompute(false) (kernel_loo_opt.hpp:77-95), e = 1e-4."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(N, D))
    Y = np.stack([np.cos(X.sum(axis=1) * (p + 1)) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Y)
    nt = D + 1 if kind == O.SE_ARD else 2
    h = new_gp(lib, kind, X, om, np.zeros(nt), 0.01)
    tot = 0.0
    e = 1e-4
    for _ in range(trials):
        th = rng.uniform(-1.0, 1.0, size=nt + (1 if optimize_noise else 0))

        def loo(t):
            h.set_kernel(kind, t[:nt], np.exp(2 * t[nt]) if optimize_noise else 0.01)
            assert h.compute() == 0
            return h.log_loo_cv()

        loo(th)
        g = h.log_loo_cv_grad(optimize_noise)
        fd = np.zeros_like(th)
        for j in range(th.size):
            tp, tm = th.copy(), th.copy()
            tp[j] += e
            tm[j] -= e
            fd[j] = (loo(tp) - loo(tm)) / (2 * e)
        tot += np.linalg.norm(fd - g) / max(1.0, np.linalg.norm(g))
    assert tot < trials * 1e-4
    h.close()


def check_update_alpha_and_clone(lib):
    """recompute(true,false) (gp.hpp:241-252) == fresh compute on the new observations;
    clone has value semantics (kernel_lf_opt.hpp:79)."""
    rng = np.random.default_rng(9)
    X = rng.uniform(-1, 1, size=(37, 3))
    Y1 = rng.normal(size=(37, 2))
    Y2 = rng.normal(size=(37, 2))
    om1, _ = O.obs_mean_data(Y1)
    om2, _ = O.obs_mean_data(Y2)
    h = new_gp(lib, O.MATERN52, X, om1, np.array([0.1, -0.2]), 0.02)
    h.compute()
    c = h.clone()
    h.update_alpha(om2)
    f = new_gp(lib, O.MATERN52, X, om2, np.array([0.1, -0.2]), 0.02)
    f.compute()
    assert np.allclose(h.get_alpha(), f.get_alpha(), rtol=1e-10, atol=1e-12)
    assert abs(h.log_lik() - f.log_lik()) < 1e-10 * abs(f.log_lik())
    # the clone still answers for Y1
    f1 = new_gp(lib, O.MATERN52, X, om1, np.array([0.1, -0.2]), 0.02)
    f1.compute()
    assert np.allclose(c.get_alpha(), f1.get_alpha(), rtol=1e-10, atol=1e-12)
    assert abs(c.log_lik() - f1.log_lik()) < 1e-10 * abs(f1.log_lik())
    for x in (h, c, f, f1):
        x.close()


def check_host_K(lib):
    """HOST_K fallback: K built by the caller (user kernel functor) == device-built K path."""
    rng = np.random.default_rng(13)
    X = rng.uniform(-1, 1, size=(50, 3))
    Y = rng.normal(size=(50, 1))
    om, _ = O.obs_mean_data(Y)
    th = np.array([0.2, -0.1, 0.3, 0.1])
    a = new_gp(lib, O.SE_ARD, X, om, th, 0.01)
    a.compute()
    K = O.kernel_matrix(O.SE_ARD, X, th, 0.01)
    b = new_gp(lib, 4, X, om, th, 0.01)
    b.set_K_host(K)
    b.compute()
    assert np.allclose(a.get_L(), b.get_L(), rtol=1e-10, atol=1e-12)
    assert abs(a.log_lik() - b.log_lik()) < 1e-10 * abs(a.log_lik())
    a.close()
    b.close()


def check_not_pd(lib):
    """The reference never checks LLT::info() (gp.hpp:565); the C-ABI reports the first
    non-positive pivot (1-based) instead of silently producing NaNs."""
    X = np.zeros((5, 2))
    X[3] = 1.0
    om = np.ones((5, 1))
    K = np.eye(5)
    K[2, 2] = -1.0
    h = new_gp(lib, 4, X, om, np.zeros(3), 0.01)
    h.set_K_host(K)
    assert h.compute() == 3
    h.close()
