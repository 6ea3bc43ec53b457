#!/usr/bin/env python
"""Row X1 of the round-1 verdict / BASELINE configs[4] "bf16 kernel matrix + fp64 factor" / north_star "MFMA fp64/bf16
tiles": what does bf16 do to this path's RESULTS?  A numerics study by emulation (numpy, CPU): the blocked
right-looking Cholesky of csrc (256-column outer panels) with

  A  the trailing-update operands rounded to bf16 (8-bit significand), products accumulated in fp64 — what a bf16 MFMA
     trailing update would compute at best;
  B  the same with a 3-way bf16 split of the operands (x = b1 + b2 + b3, 6 of the 9 cross products kept: ~24 bits);
  C  the kernel matrix itself stored in bf16, factor in fp64 (config 5 as literally worded);

each with and without one step of iterative refinement of alpha against the fp64 K, measured against the all-fp64
factorisation: L L^T residual, alpha, log-lik, mu and sigma^2 (incl. + noise) at 256 query points.  The bar of
BASELINE.json is 1e-8 relative on mu and sigma^2.  Prints one JSON object (committed as profiles/r02_bf16_study.json).
"""
import json
import sys
from pathlib import Path

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import synth  # noqa: E402


def bf16(x):
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32).astype(np.float64)


def split3(x):
    b1 = bf16(x)
    b2 = bf16(x - b1)
    b3 = bf16(x - b1 - b2)
    return b1, b2, b3


def chol_blocked(K, nb=256, mode="fp64"):
    """right-looking, panels of nb columns; mode: how the trailing update's operands are represented"""
    A = K.copy()
    N = A.shape[0]
    for p0 in range(0, N, nb):
        pe = min(p0 + nb, N)
        A[p0:pe, p0:pe] = sla.cholesky(A[p0:pe, p0:pe], lower=True, check_finite=False)
        if pe < N:
            A[pe:, p0:pe] = sla.solve_triangular(A[p0:pe, p0:pe], A[pe:, p0:pe].T, lower=True, check_finite=False).T
            L21 = A[pe:, p0:pe]
            if mode == "fp64":
                U = L21 @ L21.T
            elif mode == "bf16":
                q = bf16(L21)
                U = q @ q.T
            else:  # bf16x3: 6 products (terms below 2^-24 of the leading one dropped)
                b1, b2, b3 = split3(L21)
                U = b1 @ b1.T + (b1 @ b2.T + b2 @ b1.T) + (b2 @ b2.T + b1 @ b3.T + b3 @ b1.T)
            A[pe:, pe:] -= U
    return np.tril(A)


def se_ard(X1, X2):
    sq1, sq2 = (X1 * X1).sum(1), (X2 * X2).sum(1)
    return np.exp(-0.5 * np.maximum(sq1[:, None] + sq2[None, :] - 2.0 * X1 @ X2.T, 0.0))


def study(N, noise, seed=0):
    X, Y = synth.make_problem("c2", N=N, seed=20260927 + seed)
    om, mean = synth.obs_mean_data(Y)
    K = se_ard(X, X)
    K[np.diag_indices(N)] += noise + 1e-8
    rng = np.random.default_rng(1)
    Xq = rng.uniform(0, 1, size=(256, 6))
    Ks = se_ard(X, Xq)

    def post(L, refine=False, Kfac=None):
        a = sla.cho_solve((L, True), om, check_finite=False)
        if refine:  # one step against the fp64 K
            a = a + sla.cho_solve((L, True), om - K @ a, check_finite=False)
        Z = sla.solve_triangular(L, Ks, lower=True, check_finite=False)
        mu = Ks.T @ a + mean
        s2 = np.maximum(1.0 - (Z * Z).sum(0), 0.0) + noise
        ll = -0.5 * float((om * a).sum()) - float(np.log(np.diag(L)).sum()) - 0.5 * N * np.log(2 * np.pi)
        return a, mu, s2, ll

    L0 = chol_blocked(K, mode="fp64")
    a0, mu0, s20, ll0 = post(L0)
    res = {"N": N, "noise": noise, "cond_estimate": float(N / (noise + 1e-8))}
    for name, mode in (("A_bf16_trailing_update", "bf16"), ("B_bf16x3_trailing_update", "bf16x3")):
        try:
            L = chol_blocked(K, mode=mode)
        except np.linalg.LinAlgError as e:
            res[name] = {"failed": f"not positive definite: {e}"}
            continue
        out = {"LLt_residual": float(np.linalg.norm(L @ L.T - K) / np.linalg.norm(K))}
        for tag, rf in (("", False), ("_refined", True)):
            a, mu, s2, ll = post(L, refine=rf)
            out["alpha_rel" + tag] = float(np.linalg.norm(a - a0) / np.linalg.norm(a0))
            out["mu_max_rel" + tag] = float(np.max(np.abs(mu - mu0) / np.maximum(np.abs(mu0), 1e-3)))
            out["sigma2_max_rel" + tag] = float(np.max(np.abs(s2 - s20) / s20))
            out["log_lik_rel" + tag] = float(abs(ll - ll0) / abs(ll0))
        res[name] = out
    # C: the kernel matrix itself in bf16
    Kb = bf16(K)
    Kb = np.tril(Kb) + np.tril(Kb, -1).T
    try:
        Lc = sla.cholesky(Kb, lower=True, check_finite=False)
        a, mu, s2, ll = post(Lc)
        res["C_bf16_kernel_matrix"] = {"mu_max_rel": float(np.max(np.abs(mu - mu0) / np.maximum(np.abs(mu0), 1e-3))),
                                       "sigma2_max_rel": float(np.max(np.abs(s2 - s20) / s20))}
    except np.linalg.LinAlgError:
        w = np.linalg.eigvalsh(Kb)
        res["C_bf16_kernel_matrix"] = {"failed": "bf16(K) is not positive definite", "min_eigenvalue": float(w[0]),
                                       "negative_eigenvalues": int((w < 0).sum())}
    return res


def main():
    out = {"bar": "mu and sigma^2 within 1e-8 relative of the fp64 reference (BASELINE.json north_star)", "cases": []}
    for N, noise in ((200, 0.01), (200, 1e-10), (2048, 0.01), (4096, 0.01)):
        out["cases"].append(study(N, noise))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
