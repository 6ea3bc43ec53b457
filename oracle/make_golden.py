"""Mint the golden fixtures under tests/golden/.

The reference's own tests hold no golden vectors for the GP path and the
reference cannot be built here (no Eigen/Boost), so goldens are minted from
  * 50-digit mpmath evaluation of the reference's formulas (small N: ground truth), and
  * numpy/scipy LAPACK evaluation (medium N),
and the C oracle (oracle/gp_oracle.c) is pinned against both in tests/test_oracle.py.

Run:  python oracle/make_golden.py        (takes ~1-2 min; output is committed)
      python oracle/make_golden.py --only-new   (mint only the mpmath fixtures that are not there yet)
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import np_oracle as O  # noqa: E402

OUT = ROOT / "tests" / "golden"

KN = {O.SE_ARD: "se_ard", O.MATERN52: "matern52", O.MATERN32: "matern32", O.EXP: "exp"}


def small_cases():
    rng = np.random.default_rng(20260925)
    cases = []
    #          kind        N   D  P  noise  optimize_noise
    specs = [(O.SE_ARD, 24, 3, 1, 0.01, False),
             (O.SE_ARD, 40, 4, 2, 0.01, True),      # test_gp.cpp:131-271 shape (N=40, D=4, P=2)
             (O.SE_ARD, 33, 6, 1, 0.05, False),     # ragged size (not a multiple of anything)
             (O.MATERN52, 32, 2, 1, 0.01, False),
             (O.MATERN52, 48, 5, 2, 0.02, True),
             (O.MATERN32, 20, 3, 1, 0.01, False),
             (O.EXP, 20, 3, 1, 0.01, True),
             (O.SE_ARD, 1, 2, 1, 0.01, False),      # single sample
             (O.SE_ARD, 2, 1, 1, 0.01, False),
             # SE-ARD with Lambda columns (squared_exp_ard.hpp:109-126, :142-146); appended, so that the cases
             # above keep their random draws
             (O.SE_ARD, 28, 3, 1, 0.02, True, 1),
             (O.SE_ARD, 36, 5, 2, 0.01, False, 2)]
    for spec in specs:
        kind, N, D, P, noise, on = spec[:6]
        lam = spec[6] if len(spec) > 6 else 0
        X = rng.uniform(-1.5, 1.5, size=(N, D))
        Y = np.stack([np.cos(X.sum(axis=1) * (p + 1)) + 0.1 * rng.normal(size=N) for p in range(P)], axis=1)
        om, mean = O.obs_mean_data(Y)
        nt = D + D * lam + 1 if kind == O.SE_ARD else 2
        theta = rng.uniform(-0.7, 0.7, size=nt)
        Xq = rng.uniform(-1.8, 1.8, size=(7, D))
        Xq[0] = X[0]  # a query ON a training point (sigma^2 cancellation / clamp path)
        cases.append(dict(kind=kind, X=X, Y=Y, obs_mean=om, mean=mean, theta=theta, noise=noise,
                          optimize_noise=on, Xq=Xq, lam=lam))
    return cases


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    # ---- mpmath ground truth, small N
    only_new = "--only-new" in sys.argv  # mint the fixtures that do not exist yet, leave the others alone
    for i, c in enumerate(small_cases()):
        tag = KN[c["kind"]] + (f"_lambda{c['lam']}" if c["lam"] else "")
        name = f"mp_{i:02d}_{tag}_n{c['X'].shape[0]}_d{c['X'].shape[1]}_p{c['Y'].shape[1]}.npz"
        if only_new and (OUT / name).exists():
            continue
        g = O.mp_gp(c["kind"], c["X"], c["obs_mean"], c["theta"], c["noise"], Xq=c["Xq"],
                    optimize_noise=c["optimize_noise"])
        np.savez_compressed(OUT / name, kind=c["kind"], X=c["X"], Y=c["Y"], obs_mean=c["obs_mean"], mean=c["mean"],
                            theta=c["theta"], noise=c["noise"], optimize_noise=c["optimize_noise"], Xq=c["Xq"],
                            L=g["L"], alpha=g["alpha"], log_lik=g["log_lik"], grad=g["grad"],
                            kta=g["kta"], var_raw=g["var_raw"], Kinv=g["Kinv"], source="mpmath-50")
        print("wrote", name, "log_lik", g["log_lik"])
    if only_new:
        return
    # ---- LAPACK, medium N (scalars + samples only, to keep the files small)
    for cfg, kind, N, theta_scale in (("c1", O.SE_ARD, 200, 0.3), ("c2", O.SE_ARD, 512, 0.0),
                                      ("c3", O.MATERN52, 384, 0.0)):
        X, Y = O.make_problem(cfg, N=N)
        om, mean = O.obs_mean_data(Y)
        D = X.shape[1]
        nt = D + 1 if kind == O.SE_ARD else 2
        rng = np.random.default_rng(7 + N)
        theta = theta_scale * rng.uniform(-1, 1, size=nt)
        noise = 0.01
        K, L, alpha = O.gp_fit(kind, X, om, theta, noise)
        ll = O.log_lik(L, om, alpha)
        grad = O.log_lik_grad(kind, X, theta, noise, L, alpha, optimize_noise=True)
        Xq = rng.uniform(X.min(), X.max(), size=(64, D))
        kta, var_raw = O.query(kind, X, theta, L, alpha, Xq)
        idx = rng.integers(0, N, size=(256, 2))
        idx = np.sort(idx, axis=1)[:, ::-1]  # i >= j
        name = f"np_{cfg}_{KN[kind]}_n{N}.npz"
        np.savez_compressed(OUT / name, kind=kind, config=cfg, N=N, theta=theta, noise=noise, mean=mean,
                            log_lik=ll, grad=grad, Xq=Xq, kta=kta, var_raw=var_raw, alpha=alpha,
                            L_idx=idx, L_samples=L[idx[:, 0], idx[:, 1]], L_diag=np.diag(L).copy(),
                            source="numpy-scipy-lapack")
        print("wrote", name, "log_lik", ll)


if __name__ == "__main__":
    main()
