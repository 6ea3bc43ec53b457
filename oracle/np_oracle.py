"""Independent numpy/scipy (LAPACK) and mpmath restatements of limbo's GP path.

TEST INFRASTRUCTURE ONLY — imported from tests/ and oracle/make_golden.py, never by the
product.  Written independently of gp_oracle.c (vectorised / LAPACK / arbitrary precision
instead of scalar loops) so that the two can cross-validate each other; each function cites
the reference file:line (relative to /root/reference) whose semantics it follows.
"""
from __future__ import annotations

import numpy as np

SE_ARD, MATERN52, MATERN32, EXP = 0, 1, 2, 3


# --------------------------------------------------------------------------- numpy / LAPACK
def kernel_cross(kind, X1, X2, theta):
    """k(x1_i, x2_j) without noise.  squared_exp_ard.hpp:138-151 (k=0),
    matern_five_halves.hpp:104-113, matern_three_halves.hpp:101-107, exp.hpp:97-102."""
    X1 = np.asarray(X1, float)
    X2 = np.asarray(X2, float)
    theta = np.asarray(theta, float)
    D = X1.shape[1]
    if kind == SE_ARD and theta.size > D + 1:
        # k = (T - 1) / D - 1 columns of Lambda: M = Lambda Lambda^T + diag(ell^-2), squared_exp_ard.hpp:142-146
        k = (theta.size - 1) // D - 1
        A = theta[D:D + D * k].reshape(k, D).T  # _A(i, j) = p((j + 1) D + i)
        M = A @ A.T + np.diag(np.exp(theta[:D]) ** -2.0)
        d = X1[:, None, :] - X2[None, :, :]
        z = np.einsum("nma,ab,nmb->nm", d, M, d)
        return np.exp(2.0 * theta[-1]) * np.exp(-0.5 * z)
    if kind == SE_ARD:
        ell = np.exp(theta[:D])
        sf2 = np.exp(2.0 * theta[D])
        q = (X1[:, None, :] - X2[None, :, :]) / ell
        return sf2 * np.exp(-0.5 * np.sum(q * q, axis=2))
    l = np.exp(theta[0])
    sf2 = np.exp(2.0 * theta[1])
    diff = X1[:, None, :] - X2[None, :, :]
    s = np.sum(diff * diff, axis=2)
    if kind == MATERN52:
        d = np.sqrt(s)
        t1 = np.sqrt(5.0) * d / l
        t2 = 5.0 * (d * d) / (3.0 * l * l)
        return sf2 * (1 + t1 + t2) * np.exp(-t1)
    if kind == MATERN32:
        t = np.sqrt(3.0) * np.sqrt(s) / l
        return sf2 * (1 + t) * np.exp(-t)
    return sf2 * np.exp(-0.5 * s / (l * l))


def kernel_matrix(kind, X, theta, noise):
    """gp.hpp:550-562 with BaseKernel::operator() (kernel.hpp:81-84): +noise+1e-8 on i==j."""
    K = kernel_cross(kind, X, X, theta)
    K = np.tril(K) + np.tril(K, -1).T  # the reference mirrors the lower triangle
    K[np.diag_indices_from(K)] += noise + 1e-8
    return K


def kernel_grad_tensor(kind, X, theta):
    """dk/dtheta_t for all pairs: (T, N, N).  squared_exp_ard.hpp:127-135,
    matern_five_halves.hpp:115-133, matern_three_halves.hpp:109-121, exp.hpp:104-113."""
    X = np.asarray(X, float)
    theta = np.asarray(theta, float)
    N, D = X.shape
    if kind == SE_ARD and theta.size > D + 1:  # squared_exp_ard.hpp:109-126
        k = (theta.size - 1) // D - 1
        A = theta[D:D + D * k].reshape(k, D).T
        kv = kernel_cross(kind, X, X, theta)
        d = X[:, None, :] - X[None, :, :]
        G = np.empty((theta.size, N, N))
        for a in range(D):
            G[a] = (d[:, :, a] / np.exp(theta[a])) ** 2 * kv
        for j in range(k):
            proj = d @ A[:, j]
            for a in range(D):
                G[(j + 1) * D + a] = -proj * d[:, :, a] * kv
        G[-1] = 2 * kv
        return G
    if kind == SE_ARD:
        ell = np.exp(theta[:D])
        sf2 = np.exp(2.0 * theta[D])
        q = (X[:, None, :] - X[None, :, :]) / ell
        z = q * q
        k = sf2 * np.exp(-0.5 * z.sum(axis=2))
        G = np.empty((D + 1, N, N))
        for d in range(D):
            G[d] = z[:, :, d] * k
        G[D] = 2 * k
        return G
    l = np.exp(theta[0])
    sf2 = np.exp(2.0 * theta[1])
    diff = X[:, None, :] - X[None, :, :]
    s = np.sum(diff * diff, axis=2)
    G = np.empty((2, N, N))
    if kind == MATERN52:
        d = np.sqrt(s)
        t1 = np.sqrt(5.0) * d / l
        t2 = 5.0 * (d * d) / (3.0 * l * l)
        r = np.exp(-t1)
        G[0] = sf2 * (r * t1 * (1 + t1 + t2) + (-t1 - 2.0 * t2) * r)
        G[1] = 2 * sf2 * (1 + t1 + t2) * r
    elif kind == MATERN32:
        t = np.sqrt(3.0) * np.sqrt(s) / l
        r = np.exp(-t)
        G[0] = sf2 * (-t * r + (1 + t) * t * r)
        G[1] = 2 * sf2 * (1 + t) * r
    else:
        r = s / (l * l)
        k = sf2 * np.exp(-0.5 * r)
        G[0] = r * k
        G[1] = 2 * k
    return G


def gp_fit(kind, X, obs_mean, theta, noise):
    """compute(): K, L, alpha (gp.hpp:550-571, :605-611) via LAPACK dpotrf/dtrtrs."""
    import scipy.linalg as sla

    obs_mean = np.asarray(obs_mean, float)
    if obs_mean.ndim == 1:
        obs_mean = obs_mean[:, None]
    K = kernel_matrix(kind, X, theta, noise)
    L = sla.cholesky(K, lower=True)
    y = sla.solve_triangular(L, obs_mean, lower=True)
    alpha = sla.solve_triangular(L, y, lower=True, trans="T")
    return K, L, alpha


def log_lik(L, obs_mean, alpha):
    """gp.hpp:267-282 (P-quirk: logdet and n log 2pi not multiplied by P)."""
    obs_mean = np.asarray(obs_mean, float).reshape(L.shape[0], -1)
    n = L.shape[0]
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    a = np.sum(obs_mean * alpha)
    return -0.5 * a - 0.5 * logdet - 0.5 * n * np.log(2 * np.pi)


def inv_kernel(L):
    """gp.hpp:254-264."""
    import scipy.linalg as sla

    n = L.shape[0]
    Li = sla.solve_triangular(L, np.eye(n), lower=True)
    return Li.T @ Li


def log_lik_grad(kind, X, theta, noise, L, alpha, optimize_noise=False):
    """gp.hpp:285-311 + kernel.hpp:86-96: sum over i>=j with 1/2 on the diagonal."""
    Kinv = inv_kernel(L)
    W = alpha @ alpha.T - Kinv
    G = kernel_grad_tensor(kind, X, theta)
    n = L.shape[0]
    tri = np.tril(np.ones((n, n)))
    tri[np.diag_indices(n)] = 0.5
    g = [np.sum(W * tri * G[t]) for t in range(G.shape[0])]
    if optimize_noise:
        g.append(np.sum(np.diag(W) * 0.5 * 2.0 * noise))
    return np.array(g)


def query(kind, X, theta, L, alpha, Xq):
    """gp.hpp:613-632: returns (k*^T alpha [M,P], k(v,v) - ||L^-1 k*||^2 [M])."""
    import scipy.linalg as sla

    Ks = kernel_cross(kind, X, Xq, theta)  # N x M, no noise (kernel.hpp:81 defaults i=-1,j=-2)
    kta = Ks.T @ alpha
    Z = sla.solve_triangular(L, Ks, lower=True)
    kvv = np.array([kernel_cross(kind, Xq[m:m + 1], Xq[m:m + 1], theta)[0, 0] for m in range(Xq.shape[0])])
    return kta, kvv - np.sum(Z * Z, axis=0)


# --------------------------------------------------------------------------- mpmath ground truth
def mp_gp(kind, X, obs_mean, theta, noise, Xq=None, optimize_noise=False, dps=50):
    """50-digit evaluation of the same formulas (exact-arithmetic stand-in for 'what the
    reference computes up to fp64 rounding').  Small N only (O(N^3) in Python)."""
    import mpmath as mp

    mp.mp.dps = dps
    X = np.asarray(X, float)
    obs_mean = np.asarray(obs_mean, float).reshape(X.shape[0], -1)
    N, D = X.shape
    P = obs_mean.shape[1]
    th = [mp.mpf(float(t)) for t in theta]
    nz = mp.mpf(float(noise))

    lam = (len(th) - 1) // D - 1 if kind == SE_ARD else 0  # columns of Lambda (squared_exp_ard.hpp:94)

    def quad(a, b):
        """(a-b)^T (Lambda Lambda^T + diag(ell^-2)) (a-b), squared_exp_ard.hpp:142-146, and the pieces of its gradient"""
        d = [mp.mpf(float(a[i])) - mp.mpf(float(b[i])) for i in range(D)]
        zs = [(d[i] / mp.exp(th[i])) ** 2 for i in range(D)]
        proj = [sum(d[i] * th[(j + 1) * D + i] for i in range(D)) for j in range(lam)]
        return d, zs, proj, sum(zs) + sum(pj * pj for pj in proj)

    def kf(a, b):
        if kind == SE_ARD and lam > 0:
            return mp.exp(2 * th[-1]) * mp.exp(-quad(a, b)[3] / 2)
        if kind == SE_ARD:
            z = mp.mpf(0)
            for d in range(D):
                q = (mp.mpf(float(a[d])) - mp.mpf(float(b[d]))) / mp.exp(th[d])
                z += q * q
            return mp.exp(2 * th[D]) * mp.exp(-z / 2)
        l = mp.exp(th[0])
        sf2 = mp.exp(2 * th[1])
        s = mp.mpf(0)
        for d in range(D):
            q = mp.mpf(float(a[d])) - mp.mpf(float(b[d]))
            s += q * q
        if kind == MATERN52:
            dd = mp.sqrt(s)
            t1 = mp.sqrt(5) * dd / l
            t2 = 5 * s / (3 * l * l)
            return sf2 * (1 + t1 + t2) * mp.exp(-t1)
        if kind == MATERN32:
            t = mp.sqrt(3) * mp.sqrt(s) / l
            return sf2 * (1 + t) * mp.exp(-t)
        return sf2 * mp.exp(-s / (2 * l * l))

    def gf(a, b):
        if kind == SE_ARD and lam > 0:  # squared_exp_ard.hpp:109-126
            d, zs, proj, z = quad(a, b)
            k = mp.exp(2 * th[-1]) * mp.exp(-z / 2)
            g = [zz * k for zz in zs]
            for j in range(lam):
                g += [-proj[j] * d[i] * k for i in range(D)]
            return g + [2 * k]
        if kind == SE_ARD:
            zs = []
            for d in range(D):
                q = (mp.mpf(float(a[d])) - mp.mpf(float(b[d]))) / mp.exp(th[d])
                zs.append(q * q)
            k = mp.exp(2 * th[D]) * mp.exp(-sum(zs) / 2)
            return [z * k for z in zs] + [2 * k]
        l = mp.exp(th[0])
        sf2 = mp.exp(2 * th[1])
        s = mp.mpf(0)
        for d in range(D):
            q = mp.mpf(float(a[d])) - mp.mpf(float(b[d]))
            s += q * q
        if kind == MATERN52:
            dd = mp.sqrt(s)
            t1 = mp.sqrt(5) * dd / l
            t2 = 5 * s / (3 * l * l)
            r = mp.exp(-t1)
            return [sf2 * (r * t1 * (1 + t1 + t2) + (-t1 - 2 * t2) * r), 2 * sf2 * (1 + t1 + t2) * r]
        if kind == MATERN32:
            t = mp.sqrt(3) * mp.sqrt(s) / l
            r = mp.exp(-t)
            return [sf2 * (-t * r + (1 + t) * t * r), 2 * sf2 * (1 + t) * r]
        r = s / (l * l)
        k = sf2 * mp.exp(-r / 2)
        return [r * k, 2 * k]

    K = mp.zeros(N, N)
    for i in range(N):
        for j in range(i + 1):
            v = kf(X[i], X[j])
            if i == j:
                v += nz + mp.mpf("1e-8")
            K[i, j] = v
            K[j, i] = v
    L = mp.cholesky(K)
    B = mp.matrix(N, P)
    for i in range(N):
        for p in range(P):
            B[i, p] = mp.mpf(float(obs_mean[i, p]))
    Y = mp.lu_solve(L, B) if False else _mp_trsm(L, B, lower=True)
    A = _mp_trsm(L.T, Y, lower=False)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(N))
    a = sum(B[i, p] * A[i, p] for i in range(N) for p in range(P))
    ll = -a / 2 - logdet / 2 - mp.mpf(N) / 2 * mp.log(2 * mp.pi)
    # K^-1
    Li = _mp_trsm(L, mp.eye(N), lower=True)
    Kinv = Li.T * Li
    T = len(th) + (1 if optimize_noise else 0)
    grad = [mp.mpf(0)] * T
    for i in range(N):
        for j in range(i + 1):
            w = sum(A[i, p] * A[j, p] for p in range(P)) - Kinv[i, j]
            g = gf(X[i], X[j])
            if optimize_noise:
                g = g + [2 * nz if i == j else mp.mpf(0)]
            f = mp.mpf("0.5") if i == j else mp.mpf(1)
            for t in range(T):
                grad[t] += w * g[t] * f
    out = {
        "K": _mp2np(K), "L": _mp2np(L), "alpha": _mp2np(A), "log_lik": float(ll),
        "Kinv": _mp2np(Kinv), "grad": np.array([float(g) for g in grad]),
    }
    if Xq is not None:
        Xq = np.asarray(Xq, float)
        M = Xq.shape[0]
        kta = np.zeros((M, P))
        var = np.zeros(M)
        for m in range(M):
            ks = mp.matrix(N, 1)
            for i in range(N):
                ks[i, 0] = kf(X[i], Xq[m])
            z = _mp_trsm(L, ks, lower=True)
            for p in range(P):
                kta[m, p] = float(sum(ks[i, 0] * A[i, p] for i in range(N)))
            var[m] = float(kf(Xq[m], Xq[m]) - sum(z[i, 0] * z[i, 0] for i in range(N)))
        out["kta"] = kta
        out["var_raw"] = var
    return out


def _mp_trsm(T, B, lower=True):
    import mpmath as mp

    n = T.rows
    X = B.copy()
    rng = range(n) if lower else range(n - 1, -1, -1)
    for c in range(B.cols):
        for i in rng:
            s = X[i, c]
            ks = range(i) if lower else range(i + 1, n)
            for k in ks:
                s -= T[i, k] * X[k, c]
            X[i, c] = s / T[i, i]
    return X


def _mp2np(M):
    return np.array([[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)])


# synthetic data and the host-side finish of a query live in the product package (pure numpy)
from limbo_amd.synth import finish_query, hartmann6, make_problem, obs_mean_data, rastrigin  # noqa: E402,F401
