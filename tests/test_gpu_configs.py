"""GPU parity at the sizes and regimes BASELINE.json's configs name (run with -m gpu on an MI355X), through the C-ABI:

* C3 at full size (N=16384, D=12, Matern-5/2): mu / sigma^2 against LAPACK on 256 query points, 1e-8, at theta = 0 and at four
  points of U[-1, 1]^2; the M = 100 000 batch as ONE call (1024 sampled outputs vs LAPACK, bitwise equal to chunks of 4096);
  the Matern-5/2 log-lik gradient at N = 8192, D = 12 against a LAPACK gradient.
* C4 as written: 64 GPs of N=2048 through gpe_batch_compute, sampled members against the oracle.
* C5: the add_sample loop of src/benchmarks/limbo/bench.cpp:66-84 at its noise of 1e-10 and at 0.01, against the
  oracle AND against the reference itself (oracle/_ref), tolerances stated per quantity.
* the engine against the reference itself (limbo::model::GP compiled from /root/reference/src, oracle/_ref).
* eight host threads driving eight handles (factorisations, point queries, alpha refreshes): bitwise equal to the
  serial run; the block-by-block re-run of a one-launch sweep (forced) gives the oracle's numbers.
"""
import os
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

from limbo_amd import _capi, synth
from oracle import binding as OB
from oracle import np_oracle as O
from tests import parity_checks as PC
from tests.util import new_gp, relerr, relerr_norm

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_gpu_c3_full_size_vs_lapack(engine_lib):
    """BASELINE configs[2]: N=16384, D=12, Matern-5/2.  K built on the host (matern_five_halves.hpp:104-113 +
    kernel.hpp:83), LAPACK dpotrf / dtrtrs on all host cores, 256 of the query points: max rel. error of mu and of
    sigma^2 (incl. + noise, gp.hpp:166) <= 1e-8; log-lik <= 1e-10."""
    import scipy.linalg as sla
    from scipy.spatial.distance import cdist

    X, Y = synth.make_problem("c3")
    N, D = X.shape
    assert (N, D) == (16384, 12)
    om, mean = synth.obs_mean_data(Y)
    th, noise = np.zeros(2), 0.01
    h = new_gp(engine_lib, O.MATERN52, X, om, th, noise)
    assert h.compute() == 0
    ll = h.log_lik()
    rng = np.random.default_rng(33)
    Xq = rng.uniform(0, 1, size=(256, D))
    kta, var = h.query_batch(Xq)
    mu, s2 = synth.finish_query(kta, var, mean, noise)

    def matern52(A, B):
        d = cdist(A, B)
        t1 = np.sqrt(5.0) * d
        return (1.0 + t1 + 5.0 * d * d / 3.0) * np.exp(-t1)

    K = matern52(X, X)
    K[np.diag_indices(N)] += noise + 1e-8
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    del K
    z = sla.solve_triangular(L, om, lower=True, check_finite=False)
    alpha = sla.solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    ll_ref = O.log_lik(L, om, alpha)
    assert abs(ll - ll_ref) <= PC.TOL_LL * abs(ll_ref), (ll, ll_ref)
    Ks = matern52(X, Xq)
    Z = sla.solve_triangular(L, Ks, lower=True, check_finite=False)
    mur, s2r = synth.finish_query(Ks.T @ alpha, 1.0 - np.sum(Z * Z, axis=0), mean, noise)
    e_mu, e_s2 = relerr(mu, mur, floor=1e-3), relerr(s2, s2r)
    print(f"C3 full size vs LAPACK: mu {e_mu:.2e}  sigma^2 {e_s2:.2e}  log-lik {abs(ll - ll_ref) / abs(ll_ref):.2e}")
    assert e_mu < PC.TOL_MU and e_s2 < PC.TOL_VAR
    # the one-launch sweep for a handful of points agrees with the blocked solve of the batch
    k8, v8 = h.query_batch(Xq[:8])
    m8, s8 = synth.finish_query(k8, v8, mean, noise)
    assert relerr(m8, mur[:8], floor=1e-3) < PC.TOL_MU and relerr(s8, s2r[:8]) < PC.TOL_VAR
    # Round 6 (missing 6 i): the M = 100 000 batch bench.py times, as ONE gpe_query_batch call (gp.hpp:159-167 per
    # point, multi_gp.hpp:191-195) — 1024 sampled outputs against LAPACK at 1e-8, and every output bitwise equal to the same
    # points asked in chunks of 4096 (a point's answer must not depend on where it sits in a batch)
    M = 100_000
    Xb = np.random.default_rng(20260928).uniform(0, 1, size=(M, D))
    kb, vb = h.query_batch(Xb)
    pick = np.sort(rng.choice(M, size=1024, replace=False))
    Kp = matern52(X, Xb[pick])
    Zp = sla.solve_triangular(L, Kp, lower=True, check_finite=False)
    mup, s2p = synth.finish_query(Kp.T @ alpha, 1.0 - np.sum(Zp * Zp, axis=0), mean, noise)
    mub, s2b = synth.finish_query(kb[pick], vb[pick], mean, noise)
    e_mb, e_sb = relerr(mub, mup, floor=1e-3), relerr(s2b, s2p)
    print(f"C3, 100 000 points in one call, 1024 sampled vs LAPACK: mu {e_mb:.2e}  sigma^2 {e_sb:.2e}")
    assert e_mb < PC.TOL_MU and e_sb < PC.TOL_VAR
    for i0 in range(0, M, 4096):
        kc, vc = h.query_batch(Xb[i0:i0 + 4096])
        assert np.array_equal(kc, kb[i0:i0 + 4096]) and np.array_equal(vc, vb[i0:i0 + 4096]), i0
    h.close()


def _matern52_parts(X, theta):
    """d / l and the kernel function values of matern_five_halves.hpp:104-113 (no noise) for all pairs."""
    from scipy.spatial.distance import cdist

    ell, sf2 = np.exp(theta[0]), np.exp(2.0 * theta[1])
    t1 = cdist(X, X)
    t1 *= np.sqrt(5.0) / ell
    return t1, sf2


@pytest.mark.parametrize("theta", [(-0.71, 0.43), (0.55, -0.62), (-0.18, -0.9), (0.83, 0.27)])
def test_gpu_c3_full_size_off_theta_zero_vs_lapack(engine_lib, theta):
    """VERDICT r5, missing 6 (iii): configs[2] at four hyper-parameter points of U[-1, 1]^2 besides theta = 0 (length scales
    0.49 ... 2.3, sigma_f^2 0.17 ... 2.4: the conditioning of K moves by four orders of magnitude) — N = 16384, D = 12,
    Matern-5/2 (matern_five_halves.hpp:97-113), log-lik 1e-10, mu / sigma^2 on 256 points 1e-8 against LAPACK."""
    import scipy.linalg as sla
    from scipy.spatial.distance import cdist

    X, Y = synth.make_problem("c3")
    N, D = X.shape
    om, mean = synth.obs_mean_data(Y)
    th, noise = np.array(theta), 0.01
    h = new_gp(engine_lib, O.MATERN52, X, om, th, noise)
    assert h.compute() == 0
    ll = h.log_lik()
    Xq = np.random.default_rng(77).uniform(0, 1, size=(256, D))
    kta, var = h.query_batch(Xq)
    mu, s2 = synth.finish_query(kta, var, mean, noise)
    h.close()
    ell, sf2 = np.exp(th[0]), np.exp(2.0 * th[1])

    def matern52(A, B):
        t1 = cdist(A, B)
        t1 *= np.sqrt(5.0) / ell
        return sf2 * (1.0 + t1 + t1 * t1 / 3.0) * np.exp(-t1)

    K = matern52(X, X)
    K[np.diag_indices(N)] += noise + 1e-8
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    del K
    z = sla.solve_triangular(L, om, lower=True, check_finite=False)
    alpha = sla.solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    ll_ref = O.log_lik(L, om, alpha)
    Ks = matern52(X, Xq)
    Z = sla.solve_triangular(L, Ks, lower=True, check_finite=False)
    mur, s2r = synth.finish_query(Ks.T @ alpha, sf2 - np.sum(Z * Z, axis=0), mean, noise)
    e_ll, e_mu, e_s2 = abs(ll - ll_ref) / abs(ll_ref), relerr(mu, mur, floor=1e-3), relerr(s2, s2r)
    print(f"C3 full size at theta = {theta}: log-lik {e_ll:.2e}  mu {e_mu:.2e}  sigma^2 {e_s2:.2e}")
    assert e_ll <= PC.TOL_LL and e_mu < PC.TOL_MU and e_s2 < PC.TOL_VAR


def test_gpu_c3_matern52_gradient_vs_lapack(engine_lib):
    """VERDICT r5, missing 6 (ii): d log-lik / d theta of the Matern-5/2 kernel at configs[2]'s shape (D = 12; N = 8192 — half
    of C3's: K^-1 by LAPACK on the host is 2 N^3 flops and the full size takes minutes there) against a LAPACK gradient
    (gp.hpp:285-311 with matern_five_halves.hpp:115-133 and kernel.hpp:86-96 for the noise entry), 1e-6, optimize_noise off and
    on, at theta = 0 and off it.  SE-ARD has this check at its full size (test_gpu_c2_full_size_gradient_objective_vs_lapack)."""
    import scipy.linalg as sla

    X, Y = synth.make_problem("c3", N=8192)
    N, D = X.shape
    assert D == 12
    om, _ = synth.obs_mean_data(Y)
    noise = 0.01
    for th in (np.zeros(2), np.array([-0.4, 0.3])):
        h = new_gp(engine_lib, O.MATERN52, X, om, th, noise)
        assert h.compute() == 0
        ll = h.log_lik()
        g_off, g_on = h.log_lik_grad(False), h.log_lik_grad(True)
        h.close()
        t1, sf2 = _matern52_parts(X, th)
        r = np.exp(-t1)
        t2 = t1 * t1 / 3.0
        kf = sf2 * (1.0 + t1 + t2) * r
        K = kf.copy()
        K[np.diag_indices(N)] += noise + 1e-8
        L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        del K
        W = sla.cho_solve((L, True), np.eye(N), check_finite=False)  # K^-1 (gp.hpp:254-264)
        alpha = sla.cho_solve((L, True), om, check_finite=False)
        ll_ref = O.log_lik(L, om, alpha)
        W *= -1.0
        W += alpha @ alpha.T  # w = alpha alpha^T - K^-1 (gp.hpp:289)
        # grad(0) = sf2 (r t1 (1 + t1 + t2) - (t1 + 2 t2) r), grad(1) = 2 k; lower-triangle sum with 1/2 on the diagonal
        # (gp.hpp:293-308) = 1/2 of the full symmetric sum; the noise entry: kernel.hpp:93 (2 noise on i == j)
        g0 = sf2 * r * (t1 * (1.0 + t1 + t2) - (t1 + 2.0 * t2))
        gr = np.array([0.5 * np.sum(W * g0), 0.5 * np.sum(W * (2.0 * kf)), 0.5 * np.sum(np.diag(W)) * 2.0 * noise])
        e_ll, e_off, e_on = abs(ll - ll_ref) / abs(ll_ref), relerr_norm(g_off, gr[:2]), relerr_norm(g_on, gr)
        print(f"Matern-5/2 gradient, N = {N}, D = {D}, theta = {th}: log-lik {e_ll:.2e}  grad {e_off:.2e} (optimize_noise off) {e_on:.2e} (on)  ({g_on})")
        assert len(g_off) == 2 and len(g_on) == 3
        assert e_ll <= PC.TOL_LL and e_off < PC.TOL_GRAD and e_on < PC.TOL_GRAD


def test_gpu_c4_batch_64_vs_oracle(engine_lib, oracle_lib):
    """BASELINE configs[3]: 64 independent GPs, N=2048, D=6 (same X, 64 observation vectors and perturbed
    theta0: multi_gp.hpp:124-126 / parallel_repeater.hpp:88), all through ONE gpe_batch_compute (one launch sequence,
    gridDim.z = GP); 4 sampled members against the oracle (log-lik 1e-10, alpha 1e-8, mu / sigma^2 1e-8), all 64
    equal to one-at-a-time evaluation to rounding, the batch itself bitwise reproducible."""
    G, N = 64, 2048
    X, Y0 = synth.make_problem("c4", N=N)
    rng = np.random.default_rng(44)
    hs, oms, ths = [], [], []
    for g in range(G):
        Y = Y0 * rng.uniform(0.5, 1.5) + 0.1 * np.sin(3.0 * X[:, g % 6 : g % 6 + 1] + g)
        om, _ = synth.obs_mean_data(Y)
        th = rng.uniform(-1e-2, 1e-2, size=7) + (0.2 if g % 2 else 0.0)
        hs.append(new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01))
        oms.append(om)
        ths.append(th)
    st = _capi.batch_compute(hs)
    assert st == [0] * G
    ll = _capi.batch_log_lik(hs)
    assert np.all(np.isfinite(ll))
    Xq = rng.uniform(0, 1, size=(64, 6))
    for g in (0, 17, 42, 63):
        o = new_gp(oracle_lib, O.SE_ARD, X, oms[g], ths[g], 0.01)
        assert o.compute() == 0
        llo = o.log_lik()
        assert abs(ll[g] - llo) <= PC.TOL_LL * abs(llo)
        assert relerr_norm(hs[g].get_alpha(), o.get_alpha()) < 1e-8
        kg, vg = hs[g].query_batch(Xq)
        ko, vo = o.query_batch(Xq)
        assert relerr(kg, ko, floor=1e-3) < PC.TOL_MU
        assert relerr(vg + 0.01, vo + 0.01) < PC.TOL_VAR
        o.close()
    single = new_gp(engine_lib, O.SE_ARD, X, oms[0], ths[0], 0.01)
    for g in range(G):  # one at a time on one handle (the data-flow launch; the 64 went through the panels): equal to rounding —
        single.set_data(X, oms[g])  # two summation orders of a 2048-term factorisation differ by ~N eps sum|terms| ~ 1e-12 |log_lik|
        single.set_kernel(O.SE_ARD, ths[g], 0.01)
        assert single.compute() == 0
        assert abs(single.log_lik() - ll[g]) <= 1e-11 * abs(ll[g]), (g, single.log_lik(), ll[g])
    # round 4: a batch of EIGHT (BASELINE configs[3] per GPU) takes the data-flow launches in their batched form (k_tail_b: the
    # members' tiles interleaved in one grid; 64 members stay with the step-by-step panels, engine.hip: tail_plan), split as
    # tall launch | one update | closing launch (GPE_BATCH_TAIL_MAX, 1536) where the single handle runs one launch: equal to
    # rounding.  Below 1536 samples both run ONE launch and every tile does the same arithmetic in the same order: bitwise equal.
    assert _capi.batch_compute(hs[:8]) == [0] * 8
    ll8 = _capi.batch_log_lik(hs[:8])
    assert np.max(np.abs(ll8 - ll[:8]) / np.abs(ll[:8])) <= 1e-11
    for g in (0, 5, 7):
        single.set_data(X, oms[g])
        single.set_kernel(O.SE_ARD, ths[g], 0.01)
        assert single.compute() == 0
        Ls, Lb = np.tril(single.get_L()), np.tril(hs[g].get_L())
        assert np.max(np.abs(Ls - Lb)) <= 1e-12 * np.max(np.abs(Ls)), g
    n1 = 1024
    h1 = [new_gp(engine_lib, O.SE_ARD, X[:n1], oms[g][:n1], ths[g], 0.01) for g in range(8)]
    assert _capi.batch_compute(h1) == [0] * 8
    for g in (0, 5, 7):
        single.set_data(X[:n1], oms[g][:n1])
        single.set_kernel(O.SE_ARD, ths[g], 0.01)
        assert single.compute() == 0
        assert np.array_equal(np.tril(single.get_L()), np.tril(h1[g].get_L())), g
    for h in h1:
        h.close()
    single.close()
    assert _capi.batch_compute(hs) == [0] * G  # (back to the full batch for the reproducibility check below)
    assert np.array_equal(_capi.batch_log_lik(hs), ll)
    assert _capi.batch_compute(hs) == [0] * G  # the batched launch sequence itself is bitwise reproducible
    assert np.array_equal(_capi.batch_log_lik(hs), ll)
    for h in hs:
        h.close()


@pytest.mark.parametrize("noise", [1e-10, 0.01])
def test_gpu_c5_add_sample_loop(engine_lib, oracle_lib, noise):
    """BASELINE configs[4] / src/benchmarks/limbo/bench.cpp:66-84: Hartmann6, 10 random samples then 190
    add_sample() calls, at the benchmark's noise 1e-10 and at 0.01; the device state after every 10th sample and
    at the end against the oracle, and at the end against the reference itself where oracle/_ref is present.

    What is held.  noise 0.01: L 1e-10 (of max|L|), alpha 1e-7, mu 1e-8, sigma^2 1e-8.  noise 1e-10: cond(K) ~ 2e10 at
    n = 200, and rounds 1-4 therefore allowed L 1e-6 / mu 1e-4 / sigma^2 1e-6 here.  Round 5 measured what the three
    implementations actually do against a 60-digit truth (test_gpu_c5_noise_1e_10_vs_mpmath_truth,
    profiles/r05_parity_operating_points.log): engine, reference and restatement all sit within 1e-12 (L), 2e-11 (mu) and
    4e-14 (sigma^2, absolute; the variance itself is ~1e-5 there) of it — the SE kernel's backward error is benign.  So the
    same bars as at noise 0.01 now hold at 1e-10: L 1e-10, mu 1e-8 of the data scale, sigma^2 1e-10 absolute (relative is
    meaningless at 2e-5); alpha (cond-amplified, ~1e5 in size) is compared through K alpha = obs_mean."""
    rng = np.random.default_rng(2026)
    n0, n1, D = 10, 200, 6
    X = rng.uniform(0, 1, size=(n1, D))
    Y = synth.hartmann6(X)[:, None]
    th = np.zeros(D + 1)
    tight = noise > 1e-6
    om0, _ = synth.obs_mean_data(Y[:n0])
    g = new_gp(engine_lib, O.SE_ARD, X[:n0], om0, th, noise)
    o = new_gp(oracle_lib, O.SE_ARD, X[:n0], om0, th, noise)
    assert g.compute() == 0 and o.compute() == 0
    Xq = rng.uniform(0, 1, size=(32, D))
    worst = dict(L=0.0, mu=0.0, s2=0.0)
    for n in range(n0, n1):
        om, mean = synth.obs_mean_data(Y[: n + 1])
        assert g.add_sample(X[n], om) == 0
        assert o.add_sample(X[n], om) == 0
        if (n + 1) % 10 == 0:
            Lg, Lo = g.get_L(), o.get_L()
            worst["L"] = max(worst["L"], float(np.max(np.abs(Lg - Lo)) / np.max(np.abs(Lo))))
            kg, vg = g.query_batch(Xq)
            ko, vo = o.query_batch(Xq)
            mg, sg = synth.finish_query(kg, vg, mean, noise)
            mo, so = synth.finish_query(ko, vo, mean, noise)
            worst["mu"] = max(worst["mu"], float(np.max(np.abs(mg - mo)) / np.max(np.abs(Y))))
            worst["s2"] = max(worst["s2"], float(np.max(np.abs(sg - so) / so)) if tight else float(np.max(np.abs(sg - so))))
            for q in range(2):  # the per-point path of a BO loop's acquisition functor == the batch
                k1, v1 = g.query_batch(Xq[q : q + 1])
                m1, s1 = synth.finish_query(k1, v1, mean, noise)
                assert abs(m1[0, 0] - mg[q, 0]) <= 1e-8 * np.max(np.abs(Y))
    print(f"C5 noise {noise:g}: worst L {worst['L']:.2e}  mu {worst['mu']:.2e}  sigma^2 {worst['s2']:.2e}")
    tolL, tolmu, tols2 = (1e-10, 1e-8, 1e-8) if tight else (1e-10, 1e-8, 1e-10)
    assert worst["L"] <= tolL and worst["mu"] <= tolmu and worst["s2"] <= tols2, worst
    om, mean = synth.obs_mean_data(Y)
    a = g.get_alpha()
    K = O.kernel_matrix(O.SE_ARD, X, th, noise)
    assert np.linalg.norm(K @ a - om) <= (1e-9 if tight else 1e-4) * np.linalg.norm(om)
    if tight:
        assert relerr_norm(a, o.get_alpha()) < 1e-7
    # incremental == full on the device (test_gp.cpp:568-635: matrixL isApprox 1e-5)
    f = new_gp(engine_lib, O.SE_ARD, X, om, th, noise)
    assert f.compute() == 0
    assert np.max(np.abs(f.get_L() - g.get_L())) <= 1e-5 * np.max(np.abs(f.get_L()))
    f.close()
    if OB.ref_available():  # the reference's own add_sample loop
        r = OB.RefGP(O.SE_ARD, D, 1, noise=noise)
        r.set_h_params(th)
        r.compute(X[:n0], Y[:n0])
        for n in range(n0, n1):
            r.add_sample(X[n], Y[n])
        Lr = r.matrixL()
        assert np.max(np.abs(g.get_L() - Lr)) <= tolL * np.max(np.abs(Lr))
        mu_r, s2_r = r.query(Xq)
        kg, vg = g.query_batch(Xq)
        mg, sg = synth.finish_query(kg, vg, mean, noise)
        assert np.max(np.abs(mg - mu_r)) <= tolmu * np.max(np.abs(Y))
        assert (np.max(np.abs(sg - s2_r) / s2_r) if tight else np.max(np.abs(sg - s2_r))) <= tols2
    g.close()
    o.close()


@pytest.mark.skipif(not OB.ref_available(), reason="no oracle/_ref/libref.so")
@pytest.mark.parametrize("kind,N,D,P,mean_kind,on,k_lam", [
    (O.SE_ARD, 300, 6, 1, OB.MEAN_DATA, False, 0), (O.SE_ARD, 1000, 6, 2, OB.MEAN_DATA, True, 0),
    (O.MATERN52, 777, 12, 1, OB.MEAN_CONSTANT, False, 0), (O.SE_ARD, 257, 4, 1, OB.MEAN_NULL, True, 2),
    (O.MATERN32, 129, 3, 2, OB.MEAN_DATA, False, 0), (O.EXP, 64, 2, 1, OB.MEAN_DATA, True, 0)])
def test_gpu_vs_reference(engine_lib, kind, N, D, P, mean_kind, on, k_lam):
    """The HIP engine against limbo::model::GP itself (unmodified headers from /root/reference/src compiled into
    oracle/_ref/libref.so): log-lik 1e-10, L 1e-10, gradient 1e-6 (norm), mu / sigma^2 1e-8 — SURVEY §8(c)'s bars."""
    rng = np.random.default_rng(100 * kind + N)
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.sin(3 * X.sum(axis=1) + p) + 0.1 * rng.normal(size=N) for p in range(P)], axis=1)
    nt = D + D * k_lam + 1 if kind == O.SE_ARD else 2
    th = rng.uniform(-0.5, 0.5, size=nt)
    noise, const = 0.01, 0.3
    r = OB.RefGP(kind, D, P, mean=mean_kind, noise=noise, optimize_noise=on, k_lambda=k_lam, constant=const)
    r.set_h_params(np.concatenate([th, [np.log(np.sqrt(noise))]]) if on else th)
    r.compute(X, Y)
    m = {OB.MEAN_DATA: Y.mean(axis=0), OB.MEAN_NULL: np.zeros(P), OB.MEAN_CONSTANT: np.full(P, const)}[mean_kind]
    g = new_gp(engine_lib, kind, X, Y - m, th, noise)
    assert g.compute() == 0
    ll_r = r.log_lik()
    assert abs(g.log_lik() - ll_r) <= PC.TOL_LL * abs(ll_r)
    Lr = r.matrixL()
    assert np.max(np.abs(g.get_L() - Lr)) <= 1e-10 * np.max(np.abs(Lr))
    assert relerr_norm(g.get_alpha(), r.alpha()) < 1e-8
    gr = r.kernel_grad_log_lik()
    assert np.linalg.norm(g.log_lik_grad(on) - gr) <= PC.TOL_GRAD * np.linalg.norm(gr)
    Xq = np.concatenate([rng.uniform(0, 1, size=(100, D)), X[:8]])
    mu_r, s2_r = r.query(Xq)
    kta, var = g.query_batch(Xq)
    mu, s2 = synth.finish_query(kta, var, m, noise)
    assert relerr(mu, mu_r, floor=1e-3) < PC.TOL_MU and relerr(s2, s2_r) < PC.TOL_VAR
    k1, v1 = g.query_batch(Xq[:1])  # the per-point path
    m1, s1 = synth.finish_query(k1, v1, m, noise)
    assert relerr(m1, mu_r[:1], floor=1e-3) < PC.TOL_MU and relerr(s1, s2_r[:1]) < PC.TOL_VAR
    if N <= 300:  # LOO-CV value and gradient (the reference's literal N^3-per-parameter products)
        v_r = r.log_loo_cv()
        assert abs(g.log_loo_cv() - v_r) <= 1e-9 * abs(v_r)
        gl = r.kernel_grad_log_loo_cv()
        assert np.linalg.norm(g.log_loo_cv_grad(on) - gl) <= PC.TOL_GRAD * max(np.linalg.norm(gl), 1.0)
    for i in range(3):  # add_sample keeps up with the reference's incremental row
        xn, yn = rng.uniform(0, 1, size=D), rng.normal(size=P)
        r.add_sample(xn, yn)
        X, Y = np.vstack([X, xn]), np.vstack([Y, yn])
        m = {OB.MEAN_DATA: Y.mean(axis=0), OB.MEAN_NULL: np.zeros(P), OB.MEAN_CONSTANT: np.full(P, const)}[mean_kind]
        assert g.add_sample(xn, Y - m) == 0
    assert relerr_norm(g.get_alpha(), r.alpha()) < 1e-8
    assert np.max(np.abs(g.get_L() - r.matrixL())) <= 1e-10 * np.max(np.abs(Lr))
    g.close()


def _thread_workload(h, X, om, th, Xq, out, idx, rounds):
    res = []
    for it in range(rounds):
        h.set_kernel(O.SE_ARD, th + 0.01 * it, 0.01)
        assert h.compute() == 0
        res.append(h.log_lik())
        for q in range(4):  # point queries: the one-launch forward sweep
            k, v = h.query_batch(Xq[q : q + 1])
            res.append(float(k[0, 0]))
            res.append(float(v[0]))
        h.update_alpha(om[::-1].copy())  # forward + backward one-launch sweeps
        res.append(h.log_lik())
        h.update_alpha(om)
        k, v = h.query_batch(Xq)  # blocked matrix-core solve
        res.append(float(np.sum(k)))
        res.append(float(np.sum(v)))
    out[idx] = res


def test_gpu_sweeps_under_contention(engine_lib):
    """Eight host threads, eight handles of different sizes (64 .. 4096 samples, i.e. 1 .. 64 workgroups per
    sweep next to 147 KB-LDS GEMM workgroups of the other handles): factorisations, point queries and alpha
    refreshes all in flight together.  Every number must be bitwise what the same handle produces alone, and no
    sweep may have needed the block-by-block re-run."""
    sizes = [4096, 64, 2048, 700, 4096, 1500, 130, 3000]
    rng = np.random.default_rng(8)
    probs = []
    for i, N in enumerate(sizes):
        X = rng.uniform(0, 1, size=(N, 6))
        Y = synth.hartmann6(X)[:, None] + 0.05 * rng.normal(size=(N, 1))
        om, _ = synth.obs_mean_data(Y)
        probs.append((X, om, rng.uniform(-0.2, 0.2, size=7), rng.uniform(0, 1, size=(40, 6))))
    rounds = 3
    serial = [None] * len(sizes)
    hs = []
    for i, (X, om, th, Xq) in enumerate(probs):
        h = _capi.Handle(engine_lib)
        h.set_data(X, om)
        hs.append(h)
        _thread_workload(h, X, om, th, Xq, serial, i, rounds)
    conc = [None] * len(sizes)
    ths = [threading.Thread(target=_thread_workload, args=(hs[i], *probs[i], conc, i, rounds)) for i in range(len(sizes))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(len(sizes)):
        assert conc[i] is not None, f"thread {i} died"
        assert conc[i] == serial[i], (i, sizes[i])
    assert sum(h.flow_retries() for h in hs) == 0
    for h in hs:
        h.close()


def test_gpu_sweep_retry_path_gives_the_same_results():
    """GPE_FLOW_FAULT=1 makes every first attempt of a one-launch sweep count as timed out, so compute, add_sample,
    update_alpha, point queries, gradients and the LOO weights all take the block-by-block re-run; the results must
    still be the oracle's (child process: the switch is read once)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from limbo_amd import _capi\n"
            "from oracle import binding as OB\n"
            "from tests.test_gpu_parity import _small_parity\n"
            "_small_parity(_capi.load_engine(), OB.load_oracle(), N=520, seed=3)\n"
            "import numpy as np\n"
            "from limbo_amd import synth\n"
            "X, Y = synth.make_problem('c2', N=300); om, _ = synth.obs_mean_data(Y)\n"
            "h = _capi.Handle(_capi.load_engine()); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)\n"
            "assert h.compute() == 0 and h.flow_retries() >= 1, h.flow_retries()\n"
            "print('child ok', h.flow_retries())\n") % str(ROOT)
    env = dict(os.environ, GPE_FLOW_FAULT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_gpu_panel_head_tiles_handed_over_or_rederived_give_the_same_factor(engine_lib):
    """A fused panel step gets the L tiles of the panel's head row blocks from the workgroups that own them (potrf.hip:
    write-through stores + a flag word per tile); a workgroup that runs out of patience re-derives them from A, which is
    also what GPE_PANEL_HANDOVER=0 selects for every workgroup.  Both orders of arithmetic are the same products, so L
    must agree to the last bit; and it is LAPACK's L to rounding.  N = 1100 covers full panels, a ragged last panel and
    steps with 3, 2 and 1 head tiles (child process: the switch is read once)."""
    import scipy.linalg as sl
    N = 1100
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7), 0.01)
    assert h.compute() == 0
    L1 = np.tril(h.get_L())
    K = h.get_K()
    h.close()
    Lref = sl.cholesky(np.tril(K) + np.tril(K, -1).T, lower=True)
    assert np.max(np.abs(L1 - Lref)) <= 1e-10 * np.max(np.abs(Lref))
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    f = out / "L_no_handover.npy"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from limbo_amd import _capi, synth\n"
            "X, Y = synth.make_problem('c2', N=%d); om, _ = synth.obs_mean_data(Y)\n"
            "h = _capi.Handle(_capi.load_engine()); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)\n"
            "assert h.compute() == 0\n"
            "np.save(%r, np.tril(h.get_L()))\n"
            "print('child ok')\n") % (str(ROOT), N, str(f))
    Ls = {}
    for tag, extra in (("no_handover", {"GPE_PANEL_HANDOVER": "0"}), ("handover_steps", {"GPE_PANEL256": "0", "GPE_TAIL_MAX": "0"}),
                       ("handover_panel256", {"GPE_TAIL_MAX": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        Ls[tag] = np.load(f)
    # handed over or re-derived: the same products in the same order, bitwise the same factor
    assert np.array_equal(Ls["no_handover"], Ls["handover_steps"]), float(np.max(np.abs(Ls["no_handover"] - Ls["handover_steps"])))
    # k_panel256 (the whole panel as one data-flow launch) solves the factoring strips' tiles in the half-block form of the
    # inverse and sums the next diagonal block's pieces in one accumulator; the default (N = 1100 <= 2816: the whole matrix as
    # ONE k_tail launch) groups the same sums by 64-column tile: another order of the same arithmetic
    assert np.max(np.abs(Ls["no_handover"] - Ls["handover_panel256"])) <= 1e-13 * np.max(np.abs(L1))
    assert np.max(np.abs(Ls["no_handover"] - L1)) <= 1e-13 * np.max(np.abs(L1))


def test_gpu_one_launch_panels_polled_buffers_across_handles(engine_lib, monkeypatch):
    """(GPE_TAIL_MAX=0: panels to the end — under the defaults every N <= 2816 is one k_tail launch and never reaches
    k_panel256.)  k_panel256 hands block inverses and head tiles over through buffers that must hold an all-ones pattern when a launch
    starts: every launch arms the other buffer of the handle's pair, both are armed when a handle is created — also when its
    streams and scratch block come out of the pool of destroyed handles, in whatever state the last launch left them.  An odd
    and an even number of one-launch panels per evaluation (N = 1100: 4, N = 1400: 5), handles destroyed and re-created in
    between and two handles taking turns: every evaluation must reproduce the first one bit for bit."""
    monkeypatch.setenv("GPE_TAIL_MAX", "0")
    ref = {}
    for rnd in range(3):
        for N in (1100, 1400, 1100):
            X, Y = synth.make_problem("c2", N=N)
            om, _ = synth.obs_mean_data(Y)
            h = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7), 0.01)
            for _ in range(1 + rnd):
                assert h.compute() == 0 and h.flow_retries() == 0
                ll = h.log_lik()
                assert ref.setdefault(N, ll) == ll
            h.close()
    hs = []
    for N in (1100, 1400):
        X, Y = synth.make_problem("c2", N=N)
        om, _ = synth.obs_mean_data(Y)
        hs.append((N, new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7), 0.01)))
    for _ in range(3):
        for N, h in hs:
            assert h.compute() == 0 and h.log_lik() == ref[N]
    for _, h in hs:
        h.close()


@pytest.mark.parametrize("N,tail", [(1100, "0"), (1152, None), (4096, None)])
def test_gpu_panel_hand_over_timeout_is_answered_by_a_full_rerun(engine_lib, monkeypatch, N, tail):
    """GPE_HANDOVER_FAULT=1: no head workgroup publishes and every consumer gives up after a few polls, i.e. every fused
    panel step of the first attempt reports a lost hand-over.  The host must notice (info word 2), run the evaluation again
    from K on without the hand-over, count it in flow_retries(), and return the same log-likelihood as an undisturbed
    process (child process: the switch is read once).  N = 1100 with GPE_TAIL_MAX=0: the one-launch panels (k_panel256);
    N = 1152 = 18 x 64: the whole factorisation as one tiled data-flow launch (k_tail), whose block inverses are muted the same
    way; N = 4096: the tall launch + update + closing launch of round 4."""
    if tail is not None:
        monkeypatch.setenv("GPE_TAIL_MAX", tail)
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    h = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7), 0.01)
    assert h.compute() == 0 and h.flow_retries() == 0
    ll = h.log_lik()
    h.close()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from limbo_amd import _capi, synth\n"
            "X, Y = synth.make_problem('c2', N=%d); om, _ = synth.obs_mean_data(Y)\n"
            "h = _capi.Handle(_capi.load_engine()); h.set_data(X, om); h.set_kernel(0, np.zeros(7), 0.01)\n"
            "assert h.compute() == 0\n"
            "assert h.flow_retries() == 1, h.flow_retries()\n"
            "assert h.compute() == 0 and h.flow_retries() == 1  # the handle stays without the hand-over\n"
            "print('child ok %%.17g' %% h.log_lik())\n") % (str(ROOT), N)
    env = dict(os.environ, GPE_HANDOVER_FAULT="1")  # (GPE_TAIL_MAX: inherited from the monkeypatched environment)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # the re-run factorises without the hand-over (step-by-step panels): the same factor to rounding (N eps ~ 5e-13 at N = 4096; the
    # data-flow launches keep a diagonal block's updates in ONE accumulator chain since round 5, the panels subtract them in two
    # pieces: 1.0e-12 apart at N = 4096, with LAPACK's value between them)
    assert abs(float(r.stdout.split("child ok")[1]) - ll) <= 3e-12 * abs(ll)


@pytest.mark.parametrize("N,P,tail,tall", [(128, 1, None, None), (320, 2, None, None), (704, 3, None, None), (1344, 1, None, None),
                                           (1344, 2, "512", "0"), (1792, 1, "768", "0"), (2624, 1, None, "0"), (150, 1, None, None),
                                           (1100, 3, None, None), (1407, 2, "512", "0"), (2500, 1, None, None),
                                           # round 4: a tall data-flow launch + ONE update in front of the closing launch
                                           (2624, 1, None, None),      # tall 0..256 (4 tile columns x 41 row strips), closing 37
                                           (1344, 2, "512", None),     # tall 0..1024, closing 5 tile columns, two outputs
                                           (2000, 3, "512", "256"),    # (tall only from column 0: here panels to 1536, closing 7, ragged, P = 3)
                                           (3200, 1, "1024", "2304"),  # tall 0..2304 (36 x 50 row strips), closing 14
                                           (3000, 2, "1280", "1536"),  # tall 0..1792 (28 x 46 strips + ragged/rhs strip), closing 18
                                           # round 6: the default three-launch orders (N64 3392 .. 4352) at ragged sizes — the ragged block's
                                           # columns stay out of the one update (they take all their columns in their own, behind the closing
                                           # launch), its rows and the right-hand sides are that update's FMA rows or a tile row
                                           (3400, 2, None, None),      # tall 0..768, update 21 tile columns, 8 ragged rows + 2 outputs
                                           (3600, 1, None, None),      # 16 ragged rows
                                           (4100, 1, None, None),      # 4 ragged rows + 1: FMAs in front of 253 tiles
                                           (4159, 3, None, None),      # 63 ragged rows + 3 outputs
                                           (4352, 1, None, None),      # the largest three-launch order
                                           ])
def test_gpu_tiled_tail_factorisation_vs_lapack(engine_lib, monkeypatch, N, P, tail, tall):
    """k_tail: the last <= 2816 columns (GPE_TAIL_MAX; all of them when N is no larger) of a factorisation whose order is a
    multiple of 64 are factored by ONE launch, a workgroup per 64 x 64 tile, operands polled between them.  Against LAPACK on the
    host: L to 1e-10 of max|L|, alpha (the forward substitution rides along as the right-hand-side strip, P rows) to 1e-7,
    log-lik to 1e-10.  Sizes: a single tile column pair (128), tails that are the whole matrix, tails behind one-launch panels
    (1344 with a 512 tail: panels to 1024, then 5 tile columns; 1792 with 768; 2624 = 41 x 64: one panel, 37 tile columns, more
    tiles than CUs), one to three outputs; orders that are not multiples of 64 (150, 1100, 1407 = 21 x 64 + 63, 2500): the ragged
    last block and the right-hand-side rows ride in the launch as one more row strip and the panel code finishes them.  Two
    evaluations: the second one runs on the other pair of polled buffers.  Round 4 (`tall`, GPE_TALL): the columns in front of
    the closing launch as ONE tall launch of the same kernel (every row strip below rides along) + one update with k = its
    width, behind 256-column panels or from column 0 on.  (K itself is the engine's: it is held to the oracle elsewhere.)"""
    import scipy.linalg as sl
    if tail is not None:
        monkeypatch.setenv("GPE_TAIL_MAX", tail)
    if tall is not None:
        monkeypatch.setenv("GPE_TALL", tall)
    rng = np.random.default_rng(N + P)
    X = rng.uniform(0, 1, size=(N, 4))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.05 * rng.normal(size=(N, P))
    om, _ = O.obs_mean_data(Y)
    h = new_gp(engine_lib, O.SE_ARD, X, om, rng.uniform(-0.3, 0.2, size=5), 0.01)
    for rep in range(2):
        assert h.compute() == 0 and h.flow_retries() == 0
        L = np.tril(h.get_L())
        K = h.get_K()
        Ks = np.tril(K) + np.tril(K, -1).T
        Lref = sl.cholesky(Ks, lower=True)
        assert np.max(np.abs(L - Lref)) <= 1e-10 * np.max(np.abs(Lref))
        aref = sl.cho_solve((Lref, True), om)
        assert relerr_norm(h.get_alpha(), aref) < 1e-7
        # gp.hpp:267-282: log det and n log 2 pi are NOT multiplied by the number of outputs
        llref = -0.5 * np.sum(om * aref) - np.sum(np.log(np.diag(Lref))) - 0.5 * N * np.log(2 * np.pi)
        assert abs(h.log_lik() - llref) <= 1e-10 * abs(llref)
        if rep == 0:
            L0, ll0 = L, h.log_lik()
        else:
            assert np.array_equal(L, L0) and h.log_lik() == ll0
    h.close()


def test_gpu_ragged_orders_of_the_three_launch_range_vs_lapack(engine_lib):
    """Round 6: orders around the three-launch range (N64 3392 .. 4352) that are NOT multiples of 64, one to three outputs — the
    ragged block's columns stay out of the one trailing update, its rows (1 .. 63) and the right-hand sides ride under it as FMA
    rows (<= 4), as a small matrix-core product of every workgroup's column share (more: gemm_glds64.h, gemm_rhs_rows_mfma) or as a
    tile row (where that does not cost a round of the chip); the tiles of the update come from the host-built table (gemm.hip:
    tri_tile_map).  Every size: L against LAPACK 1e-10 of max|L|, alpha 1e-7, no re-run."""
    import scipy.linalg as sl
    rng = np.random.default_rng(5)
    for N in (3393, 3455, 3519, 3585, 3841, 4033, 4095, 4097, 4130, 4223, 4289, 4351, 4353):
        P = int(rng.integers(1, 4))
        X = rng.uniform(0, 1, size=(N, 4))
        Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.05 * rng.normal(size=(N, P))
        om, _ = O.obs_mean_data(Y)
        h = new_gp(engine_lib, O.SE_ARD, X, om, rng.uniform(-0.3, 0.2, size=5), 0.01)
        assert h.compute() == 0 and h.flow_retries() == 0, N
        L = np.tril(h.get_L())
        K = h.get_K()
        Lref = sl.cholesky(np.tril(K) + np.tril(K, -1).T, lower=True)
        assert np.max(np.abs(L - Lref)) <= 1e-10 * np.max(np.abs(Lref)), N
        aref = sl.cho_solve((Lref, True), om)
        assert relerr_norm(h.get_alpha(), aref) < 1e-7, N
        h.close()


@pytest.mark.parametrize("threads,per", [(4, 6), (8, 200)])
def test_gpu_concurrent_handles_do_not_starve_each_other(engine_lib, threads, per):
    """Round 4: data-flow launches wait inside the launch for lower-numbered workgroups, which is deadlock-free for ONE such
    launch at a time only — four fresh handles evaluated from four host threads filled every XCD with each other's waiting
    workgroups, ran into the bounded polls and the re-run path (1 evaluation/s).  csrc/dev.h, FlowGate: a data-flow launch
    waits for the device's previous one when that went to another stream.  Round 5 (engine.hip: ChainScope): while another
    chain is in flight an evaluation runs on one of two CU-masked streams, half of every XCD's CUs each — two chains side by
    side, each launch's lowest unfinished workgroup always resident in its own half (the reference runs independent GPs
    truly in parallel: multi_gp.hpp:124-126, parallel_repeater.hpp:86-103).  4 threads x 6 and 8 threads x 200 evaluations:
    no re-run, every log-likelihood bitwise the handle's own sequential one, and not slower than 100 evaluations/s
    (measured: 1120/s with four or eight in flight against 795 chain behind chain, profiles/r05_concurrent_chains.log)."""
    import threading
    import time
    X, Y = synth.make_problem("c2", N=4096)
    om, _ = synth.obs_mean_data(Y)
    hs = [new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7) + 1e-3 * r, 0.01) for r in range(threads)]
    ref = []
    for h in hs:
        assert h.compute() == 0
        ref.append(h.log_lik())
    got = [[] for _ in hs]

    def worker(i):
        for _ in range(per):
            assert hs[i].compute() == 0
            got[i].append(hs[i].log_lik())

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(hs))]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    for i, h in enumerate(hs):
        assert h.handover_reruns() == 0 and h.flow_retries() == 0, i
        assert len(got[i]) == per and all(v == ref[i] for v in got[i]), i
        h.close()
    print(f"{threads} handles in flight: {threads * per / dt:.0f} evaluations/s")
    assert threads * per / dt > 100.0, dt


def test_gpu_chain_partitions_give_up_when_a_mask_is_not_honoured():
    """Round 5 safety net of the CU-masked chain partitions (engine.hip: ChainScope).  They rest on ONE assumption — a launch
    on a masked stream stays inside its half of every XCD — which is checked when the streams are created (a probe launch per
    stream: disjoint sets of at most 128 places, all eight XCDs) and AGAIN at the head of every masked chain (64 workgroups
    look up where they sit).  GPE_PARTITION_FAULT=1 makes that check claim the other half: the first masked chain reports a
    violation, the process goes back to one chain at a time (a line on stderr), every result stays bitwise the sequential one."""
    code = ("import sys, threading; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from limbo_amd import _capi, synth\n"
            "eng = _capi.load_engine()\n"
            "X, Y = synth.make_problem('c2', N=2048); om, _ = synth.obs_mean_data(Y)\n"
            "hs, ref = [], []\n"
            "for r in range(3):\n"
            "    h = _capi.Handle(eng); h.set_kernel(0, np.zeros(7) + 1e-3 * r, 0.01); h.set_data(X, om); assert h.compute() == 0; ref.append(h.log_lik()); hs.append(h)\n"
            "bad = [0]\n"
            "def work(i):\n"
            "    for _ in range(40):\n"
            "        if hs[i].compute() != 0 or hs[i].log_lik() != ref[i]: bad[0] += 1\n"
            "ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]\n"
            "[t.start() for t in ts]; [t.join() for t in ts]\n"
            "assert bad[0] == 0 and all(h.flow_retries() == 0 for h in hs)\n"
            "print('child ok')\n") % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPE_PARTITION_FAULT="1"), capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr
    assert "a CU mask was not honoured" in r.stderr and "one chain at a time" in r.stderr, r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout and "limbo_amd:" not in r.stderr, r.stdout + r.stderr


def test_gpu_data_flow_buffers_across_sizes_on_one_handle(engine_lib):
    """the data-flow launches (k_tail) hand tiles over through buffers that must hold an all-ones pattern
    where the launch polls, and a launch only re-arms ITS OWN slot layout of the other buffer.  One handle whose N and P change
    between evaluations — a BO run growing past the closing launch's width (2816 -> 2880 -> 2944 changes the tile-column count
    44 -> 41 -> 42), a P change, sizes with and without a tall launch in front — must give LAPACK's factor every time: the
    engine puts the pair back to all-ones when the layout differs from the previous launch's (engine.hip: prepare_tail)."""
    import scipy.linalg as sl
    rng = np.random.default_rng(7)
    h = _capi.Handle(engine_lib)
    for N, P in [(2816, 1), (2880, 1), (2944, 1), (2880, 2), (2816, 1), (4096, 1), (3000, 1), (4096, 2), (2944, 1), (2944, 1)]:
        X = rng.uniform(0, 1, size=(N, 4))
        Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.05 * rng.normal(size=(N, P))
        om, _ = O.obs_mean_data(Y)
        h.set_data(X, om)
        h.set_kernel(int(O.SE_ARD), rng.uniform(-0.3, 0.2, size=5), 0.01)
        assert h.compute() == 0 and h.flow_retries() == 0, (N, P)
        L = np.tril(h.get_L())
        K = h.get_K()
        Lref = sl.cholesky(np.tril(K) + np.tril(K, -1).T, lower=True)
        assert np.max(np.abs(L - Lref)) <= 1e-10 * np.max(np.abs(Lref)), (N, P)
        assert relerr_norm(h.get_alpha(), sl.cho_solve((Lref, True), om)) < 1e-7, (N, P)
    h.close()


def test_gpu_cached_first_look_never_sees_a_stale_slot(engine_lib):
    """the first look at a polled hand-over slot is a cacheable (workgroup-scope) load (potrf.hip: POLL_CACHED):
    sound as long as every kernel boundary writes back / invalidates the L2 lines of the slots, so that a word which is not
    the all-ones pattern is this launch's final value — never a leftover of the launch that used the same buffer TWO launches
    ago.  One handle at N = 4096, hyper-parameters cycling through three settings (so the launch two evaluations back
    always had OTHER values in every slot), 12 evaluations back to back: every log-likelihood and every 16th factor must be
    bitwise what a fresh handle (clean buffers) gives for that setting."""
    X, Y = synth.make_problem("c2", N=4096)
    om, _ = synth.obs_mean_data(Y)
    thetas = [np.zeros(7), np.full(7, 0.15), np.linspace(-0.2, 0.2, 7)]
    fresh = []
    for th in thetas:
        f = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
        assert f.compute() == 0
        fresh.append((f.log_lik(), f.get_L()[::16, ::16].copy()))
        f.close()
    h = new_gp(engine_lib, O.SE_ARD, X, om, thetas[0], 0.01)
    for k in range(12):
        i = k % 3
        h.set_kernel(O.SE_ARD, thetas[i], 0.01)
        assert h.compute() == 0
        assert h.log_lik() == fresh[i][0], (k, i)
        if k >= 9:
            assert np.array_equal(h.get_L()[::16, ::16], fresh[i][1]), (k, i)
    assert h.flow_retries() == 0 and h.handover_reruns() == 0
    h.close()


def test_gpu_tall_pair_survives_a_batch_without_a_tall_launch(engine_lib):
    """ADVICE r4 (medium).  Two handles at N = 4096 whose tall hand-over pairs are armed with DIFFERENT parities (one and two
    single evaluations), then batched together at a size whose plan has no tall launch (N = 1024), then evaluated singly at
    N = 4096 again.  Before the fix the batch copied member 0's tall count / layout onto the other member without looking at
    its pair: its next tall launch polled a buffer still holding the previous launch's tiles (not the all-ones pattern) and
    accepted them — a silently wrong factor.  Now a pair that no launch of the batch touches keeps its own state: both handles
    give LAPACK's factor, no re-run."""
    import scipy.linalg as sl
    rng = np.random.default_rng(55)
    D = 4

    def data(N, seed):
        r = np.random.default_rng(seed)
        X = r.uniform(0, 1, size=(N, D))
        Y = np.cos(X.sum(axis=1))[:, None] + 0.05 * r.normal(size=(N, 1))
        return X, O.obs_mean_data(Y)[0]

    hs = [_capi.Handle(engine_lib) for _ in range(2)]
    th = rng.uniform(-0.3, 0.2, size=D + 1)
    big = [data(4096, 1), data(4096, 2)]
    for q, h in enumerate(hs):
        h.set_data(*big[q])
        h.set_kernel(int(O.SE_ARD), th, 0.01)
    assert hs[0].compute() == 0 and hs[0].compute() == 0  # member 0: two tall launches
    assert hs[1].compute() == 0                            # member 1: one -> the other parity
    for q, h in enumerate(hs):
        h.set_data(*data(1024, 10 + q))
        h.set_kernel(int(O.SE_ARD), th, 0.01)
    assert _capi.batch_compute(hs) == [0, 0]
    for q, h in enumerate(hs):
        X, om = data(4096, 20 + q)
        h.set_data(X, om)
        h.set_kernel(int(O.SE_ARD), th + 0.05, 0.01)
        assert h.compute() == 0 and h.flow_retries() == 0, q
        K = h.get_K()
        Lref = sl.cholesky(np.tril(K) + np.tril(K, -1).T, lower=True)
        assert np.max(np.abs(np.tril(h.get_L()) - Lref)) <= 1e-10 * np.max(np.abs(Lref)), q
        assert relerr_norm(h.get_alpha(), sl.cho_solve((Lref, True), om)) < 1e-7, q
        h.close()


@pytest.mark.parametrize("kind,D,P,lam", [(O.SE_ARD, 6, 1, 0), (O.MATERN52, 3, 2, 0), (O.SE_ARD, 4, 3, 1), (O.EXP, 2, 1, 0)])
def test_gpu_small_path_vs_oracle(engine_lib, oracle_lib, kind, D, P, lam):
    """The one-launch small-N path (csrc/small.hip: add_sample and point queries below 256 samples) across every
    block boundary (63/64/65, 127.., 255/256/257: the last call it serves and the first the general path takes
    back), P = 1..3, a Lambda column: L, alpha, log-lik, mu and sigma^2 against the oracle after every step."""
    rng = np.random.default_rng(900 + 10 * kind + P)
    n0, n1 = 50, 262
    X = rng.uniform(0, 1, size=(n1, D))
    Y = np.stack([np.sin(3 * X.sum(axis=1) + p) + 0.05 * rng.normal(size=n1) for p in range(P)], axis=1)
    nt = D + D * lam + 1 if kind == O.SE_ARD else 2
    th = rng.uniform(-0.4, 0.4, size=nt)
    noise = 0.01
    om0, _ = synth.obs_mean_data(Y[:n0])
    g = new_gp(engine_lib, kind, X[:n0], om0, th, noise)
    o = new_gp(oracle_lib, kind, X[:n0], om0, th, noise)
    assert g.compute() == 0 and o.compute() == 0
    Xq = rng.uniform(0, 1, size=(8, D))
    watch = {62, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 254, 255, 256, 257, 258, 261}
    for n in range(n0, n1):
        om, mean = synth.obs_mean_data(Y[: n + 1])
        assert g.add_sample(X[n], om) == 0
        assert o.add_sample(X[n], om) == 0
        if n in watch:
            llg, llo = g.log_lik(), o.log_lik()
            assert abs(llg - llo) <= PC.TOL_LL * abs(llo), n
            Lo = o.get_L()
            assert np.max(np.abs(g.get_L() - Lo)) <= 1e-10 * np.max(np.abs(Lo)), n
            assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-8, n
            for M in (1, 3, 8):
                kg, vg = g.query_batch(Xq[:M])
                ko, vo = o.query_batch(Xq[:M])
                mg, sg = synth.finish_query(kg, vg, mean, noise)
                mo, so = synth.finish_query(ko, vo, mean, noise)
                assert relerr(mg, mo, floor=1e-3) < PC.TOL_MU and relerr(sg, so) < PC.TOL_VAR, (n, M)
            k1, _ = g.query_batch(Xq[:2], want_var=False)  # mu only / sigma^2 only
            _, v1 = g.query_batch(Xq[:2], want_mu=False)
            kg, vg = g.query_batch(Xq[:2])
            assert np.array_equal(k1, kg) and np.array_equal(v1, vg)
    if os.environ.get("GPE_SMALL", "1") != "0":
        assert g.small_calls() > 0  # the path under test did serve these calls
    else:
        assert g.small_calls() == 0
    # recompute(., false): new observations, same factor — the small path's alpha refresh at 262 samples is the general one;
    # at 200 (a second pair of handles) it is the one-launch form
    for M in (n1, 200):
        omM, _ = synth.obs_mean_data(Y[:M])
        g2 = new_gp(engine_lib, kind, X[:M], omM, th, noise)
        o2 = new_gp(oracle_lib, kind, X[:M], omM, th, noise)
        assert g2.compute() == 0 and o2.compute() == 0
        om2 = omM[::-1].copy() * 1.5
        g2.update_alpha(om2)
        o2.update_alpha(om2)
        assert relerr_norm(g2.get_alpha(), o2.get_alpha()) < 1e-8
        assert abs(g2.log_lik() - o2.log_lik()) <= PC.TOL_LL * abs(o2.log_lik())
        g2.update_alpha()  # no new observations: the device copy
        assert relerr_norm(g2.get_alpha(), o2.get_alpha()) < 1e-8
        g2.close()
        o2.close()
    # the gradient and K^-1 work from the state the small path left (block inverses included)
    go, gg = o.log_lik_grad(False), g.log_lik_grad(False)
    assert np.linalg.norm(gg - go) <= PC.TOL_GRAD * np.linalg.norm(go)
    g.close()
    o.close()


def test_gpu_general_path_at_small_n():
    """GPE_SMALL=0: the same BO-sized loops through the general (multi-launch) path, in a child process"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from limbo_amd import _capi\n"
            "from oracle import binding as OB, np_oracle as O\n"
            "from tests import test_gpu_configs as T\n"
            "eng, orc = _capi.load_engine(), OB.load_oracle()\n"
            "T.test_gpu_small_path_vs_oracle(eng, orc, O.SE_ARD, 6, 1, 0)\n"
            "T.test_gpu_small_path_vs_oracle(eng, orc, O.SE_ARD, 4, 3, 1)\n"
            "T.test_gpu_c5_add_sample_loop(eng, orc, 0.01)\n"
            "print('child ok')\n") % str(ROOT)
    env = dict(os.environ, GPE_SMALL="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_gpu_set_L_with_lambda_columns(engine_lib):
    """load(archive, recompute=false) (gp.hpp:506-509) on an SE-ARD kernel with Lambda columns: gpe_set_L must also form the
    projections Lambda^T x of the training samples that the cross-kernel reads — a handle that never ran
    compute() answers queries exactly like the one the factor came from."""
    rng = np.random.default_rng(12)
    N, D, lam = 150, 4, 2
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.sin(3 * X.sum(axis=1))[:, None]
    om, mean = synth.obs_mean_data(Y)
    th = rng.uniform(-0.4, 0.4, size=D + D * lam + 1)
    a = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
    assert a.compute() == 0
    b = _capi.Handle(engine_lib)
    b.set_data(X, om)
    b.set_kernel(O.SE_ARD, th, 0.01)
    b.set_L(a.get_L())
    b.set_alpha(a.get_alpha())
    Xq = rng.uniform(0, 1, size=(20, D))
    for M in (1, 20):  # the small path and the batched one
        ka, va = a.query_batch(Xq[:M])
        kb, vb = b.query_batch(Xq[:M])
        assert np.max(np.abs(ka - kb)) <= 1e-12 * np.max(np.abs(ka)) and np.max(np.abs(va - vb)) <= 1e-12
    a.close()
    b.close()


def _rprop(f, init, iterations, eps_stop=0.0):
    """opt::Rprop as written (src/limbo/opt/rprop.hpp:84-144): maximises f, returns the best point SEEN"""
    delta0, dmin, dmax, em, ep = 0.1, 1e-6, 50.0, 0.5, 1.2
    params = np.array(init, float)
    delta = np.full(params.size, delta0)
    grad_old = np.zeros(params.size)
    best, best_params = -np.inf, params.copy()
    for _ in range(iterations):
        lik, g = f(params)
        if lik > best:
            best, best_params = lik, params.copy()
        grad = -g
        grad_old = grad_old * grad
        for j in range(params.size):
            if grad_old[j] > 0:
                delta[j] = min(delta[j] * ep, dmax)
            elif grad_old[j] < 0:
                delta[j] = max(delta[j] * em, dmin)
                grad[j] = 0
            params[j] += -np.sign(grad[j]) * delta[j]
        grad_old = grad
        if np.linalg.norm(grad_old) < eps_stop:
            break
    return best_params, best


@pytest.mark.parametrize("N,on", [(300, False), (700, True)])
def test_gpu_kernel_lf_opt_fit_vs_oracle_and_reference(engine_lib, oracle_lib, N, on):
    """BASELINE configs[1]'s workload end to end: a KernelLFOpt fit (model/gp/kernel_lf_opt.hpp:60-92 + opt/rprop.hpp) whose
    objective — likelihood and gradient — comes from the device (gpe_hp_objective), 30 iterations.  Rprop only looks at the
    SIGNS of the gradient, so the device-driven fit walks the same iterates as the oracle's and, at N = 300, as the
    reference's own optimize_hyperparams() on limbo::model::GP: same parameters, same likelihood."""
    rng = np.random.default_rng(31 + N)
    D = 3
    X = rng.uniform(-2, 2, size=(N, D))
    Y = (np.sin(X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2])[:, None] + 0.05 * rng.normal(size=(N, 1))
    om, _ = synth.obs_mean_data(Y)
    g = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(D + 1), 0.01)
    calls = []

    def objective(p):
        th, noise = (p[:-1], float(np.exp(2 * p[-1]))) if on else (p, 0.01)
        lik, grad, info = g.hp_objective(O.SE_ARD, th, noise, optimize_noise=on, want_grad=True)
        assert info == 0
        calls.append(lik)
        return lik, grad

    init = np.concatenate([np.zeros(D + 1), [np.log(np.sqrt(0.01))]]) if on else np.zeros(D + 1)
    th_g, ll_g = _rprop(objective, init, 30)
    o = new_gp(oracle_lib, O.SE_ARD, X, om, np.zeros(D + 1), 0.01)
    assert o.compute() == 0
    th_o, ll_o, nev = OB.kernel_lf_opt_rprop(o, optimize_noise=on, iterations=30, eps_stop=0.0)
    assert nev == 30 and len(calls) == 30
    assert np.max(np.abs(th_g - th_o)) <= 1e-9, (th_g, th_o)
    assert abs(ll_g - ll_o) <= 1e-9 * abs(ll_o)
    assert ll_g > calls[0]  # the fit improved the likelihood
    if N <= 300 and OB.ref_available():
        r = OB.RefGP(O.SE_ARD, D, 1, noise=0.01, optimize_noise=on)
        r.compute(X, Y)
        r.optimize_hyperparams(OB.OPT_KERNEL_LF, iterations=30, eps_stop=0.0)
        assert np.max(np.abs(th_g - r.h_params())) <= 1e-9
        assert abs(ll_g - r.log_lik()) <= 1e-9 * abs(ll_g)
    g.close()
    o.close()


def _lapack_grad_se_ard(X, theta, noise, om, optimize_noise):
    """K, L, K^-1, alpha, log-lik and d log-lik / d theta for SE-ARD (no Lambda) from LAPACK + numpy: gp.hpp:254-311 with
    w = alpha alpha^T - K^-1, the lower-triangle sum with 1/2 on the diagonal (:293-308) = 1/2 of the full symmetric sum;
    squared_exp_ard.hpp:127-135 for dk/dtheta, kernel.hpp:86-96 for the noise entry.  Memory-lean: one N x N at a time."""
    import scipy.linalg as sla

    N, D = X.shape
    K = O.kernel_matrix(O.SE_ARD, X, theta, noise)
    kfun = K.copy()
    kfun[np.diag_indices(N)] -= noise + 1e-8
    L = sla.cholesky(K, lower=True, check_finite=False)
    Kinv = sla.cho_solve((L, True), np.eye(N), check_finite=False)
    alpha = sla.cho_solve((L, True), om, check_finite=False)
    ll = O.log_lik(L, om, alpha)
    W = alpha @ alpha.T - Kinv
    Wk = W * kfun
    ell = np.exp(theta[:D])
    g = []
    for d in range(D):
        q = (X[:, d:d + 1] - X[:, d:d + 1].T) / ell[d]
        g.append(0.5 * np.sum(Wk * q * q))
    g.append(0.5 * np.sum(Wk) * 2.0)
    if optimize_noise:
        g.append(np.sum(np.diag(W) * 0.5 * 2.0 * noise))
    return ll, np.array(g), Kinv, alpha


@pytest.mark.parametrize("on", [False, True])
def test_gpu_c2_full_size_gradient_objective_vs_lapack(engine_lib, on):
    """BASELINE configs[1] as written — 'N=4096 D=6 SquaredExpARD + KernelLFOpt hyperparam fit': the pieces of one
    KernelLFOptimization::operator() (kernel_lf_opt.hpp:77-92) at full size against LAPACK (dpotrf / dpotrs on the
    host): K^-1 (gp.hpp:254-264) on 64 sampled columns 1e-8, d log-lik / d theta (gp.hpp:285-311) 1e-6 with and
    without the noise entry, and gpe_hp_objective at a second theta (value 1e-10, gradient 1e-6).  N=4096 is where
    the factorisation takes the look-ahead + fused-update path and K^-1 its 16-panel form — no smaller test does."""
    X, Y = synth.make_problem("c2")
    assert X.shape == (4096, 6)
    om, _ = synth.obs_mean_data(Y)
    rng = np.random.default_rng(4096 + int(on))
    th = rng.uniform(-0.3, 0.3, size=7)
    noise = 0.01
    h = new_gp(engine_lib, O.SE_ARD, X, om, th, noise)
    assert h.compute() == 0
    ll = h.log_lik()
    g = h.log_lik_grad(on)
    Kinv = h.get_Kinv()
    ll_r, g_r, Kinv_r, _ = _lapack_grad_se_ard(X, th, noise, om, on)
    cols = rng.integers(0, 4096, size=64)
    e_ll = abs(ll - ll_r) / abs(ll_r)
    e_g = relerr_norm(g, g_r)
    e_ki = relerr_norm(Kinv[:, cols], Kinv_r[:, cols])
    e_sym = np.max(np.abs(Kinv - Kinv.T))
    print(f"C2 full size vs LAPACK (optimize_noise={on}): log-lik {e_ll:.2e}  grad {e_g:.2e}  K^-1 {e_ki:.2e}")
    assert e_ll <= PC.TOL_LL and e_g < PC.TOL_GRAD and e_ki < 1e-8 and e_sym == 0.0
    # one objective evaluation at another theta, as the optimiser issues it (same X, resident buffers)
    th2 = th + rng.uniform(-0.2, 0.2, size=7)
    ll2, g2, info = h.hp_objective(O.SE_ARD, th2, noise, optimize_noise=on, want_grad=True)
    ll2_r, g2_r, _, _ = _lapack_grad_se_ard(X, th2, noise, om, on)
    print(f"   hp_objective: log-lik {abs(ll2 - ll2_r) / abs(ll2_r):.2e}  grad {relerr_norm(g2, g2_r):.2e}")
    assert info == 0 and abs(ll2 - ll2_r) <= PC.TOL_LL * abs(ll2_r) and relerr_norm(g2, g2_r) < PC.TOL_GRAD
    assert h.flow_retries() == 0
    h.close()


@pytest.mark.parametrize("N", [1025, 1700, 2944])
def test_gpu_kinv_recursion_at_ragged_orders_vs_lapack(engine_lib, N):
    """K^-1 (gp.hpp:254-264) and d log-lik / d theta (gp.hpp:285-311) where the recursion of csrc/inv2.hip meets a partial last
    panel and partial last tiles (N = 1025: one sample beyond a tile edge; 1700; 2944 = 11.5 panels): the whole matrix against
    dpotrf / dpotri-style LAPACK on the host 1e-9, the gradient 1e-6; then the same handle with 37 samples fewer (the k ranges of a
    ragged order run over the pads of the recursion's buffers, which must read as zero again) and with the samples back."""
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    rng = np.random.default_rng(N)
    th = rng.uniform(-0.3, 0.3, size=7)
    noise = 0.01
    h = new_gp(engine_lib, O.SE_ARD, X, om, th, noise)
    for Xs, oms in ((X, om), (X[:N - 37], om[:N - 37]), (X, om)):
        h.set_data(Xs, oms)
        assert h.compute() == 0
        g = h.log_lik_grad(True)
        Kinv = h.get_Kinv()
        _, g_r, Kinv_r, _ = _lapack_grad_se_ard(Xs, th, noise, oms, True)
        e_ki, e_g = relerr_norm(Kinv, Kinv_r), relerr_norm(g, g_r)
        print(f"N = {len(Xs)}: K^-1 {e_ki:.2e}  grad {e_g:.2e}")
        assert not np.isnan(Kinv).any() and e_ki < 1e-9 and e_g < PC.TOL_GRAD and np.max(np.abs(Kinv - Kinv.T)) == 0.0
    h.close()


def test_gpu_c2_full_size_fit_lockstep_vs_lapack(engine_lib):
    """BASELINE configs[1] end to end at its own size: a KernelLFOpt fit (model/gp/kernel_lf_opt.hpp:60-69 -> opt/rprop.hpp:84-144)
    at N = 4096 with 4 restarts (parallel_repeater.hpp:84-105) advanced in lock-step, every iteration ONE
    gpe_batch_hp_objective (limbo_amd/hpfit.py, the mirror of the drop-in's opt/batched_rprop.hpp).  The first 5 iterates of
    restart 0 against limbo's sequential Rprop driven by a LAPACK objective on the host (theta 1e-8, log-lik 1e-10 — Rprop
    only looks at gradient signs, so equal iterates mean every sign agreed), and the 10th iterate of restarts 0 and 3
    evaluated by LAPACK where the device fit stands (log-lik 1e-10, gradient 1e-6)."""
    from limbo_amd import hpfit

    N, G, iters = 4096, 4, 10
    X, Y = synth.make_problem("c2", N=N)
    om, _ = synth.obs_mean_data(Y)
    rng = np.random.default_rng(11)
    inits = rng.uniform(-1e-2, 1e-2, size=(G, 7))
    inits[0] = 0.0
    hs = []
    for _ in range(G):
        h = _capi.Handle(engine_lib)
        h.set_data(X, om)
        hs.append(h)
    trace = []
    bp, bl = hpfit.kernel_lf_opt_lockstep(hs, O.SE_ARD, inits, noise=0.01, optimize_noise=False, iterations=iters, trace=trace)
    assert len(trace) == iters and all(h.flow_retries() == 0 for h in hs)
    seq = []

    def lapack_objective(p):
        ll, g, _, _ = _lapack_grad_se_ard(X, p, 0.01, om, False)
        seq.append((p.copy(), ll, g))
        return ll, g

    _rprop(lapack_objective, inits[0], 5)
    for i, (p, ll, g) in enumerate(seq):
        pd, ld, gd = trace[i][0][0], trace[i][1][0], trace[i][2][0]
        assert np.max(np.abs(pd - p)) <= 1e-8, (i, pd, p)
        assert abs(ld - ll) <= PC.TOL_LL * abs(ll) and relerr_norm(gd, g) < PC.TOL_GRAD, (i, ld, ll)
    for r in (0, 3):
        pd, ld, gd = trace[iters - 1][0][r], trace[iters - 1][1][r], trace[iters - 1][2][r]
        ll, g, _, _ = _lapack_grad_se_ard(X, pd, 0.01, om, False)
        assert abs(ld - ll) <= PC.TOL_LL * abs(ll) and relerr_norm(gd, g) < PC.TOL_GRAD, (r, ld, ll)
    assert np.all(bl >= np.array([t[1] for t in trace]).max(axis=0) - 1e-12)  # best seen (rprop.hpp:115-118)
    assert bl[0] > trace[0][1][0]  # the fit improved the likelihood
    for h in hs:
        h.close()


def _lapack_posterior_se_ard(X, theta, noise, om, mean, Xq):
    """log-lik, mu, sigma^2 (gp.hpp:267-282, :613-624, :166) and cond_1(K) from LAPACK: dpotrf, dpotrs, dtrtrs, dpocon."""
    import scipy.linalg as sla
    from scipy.linalg import lapack

    K = O.kernel_matrix(O.SE_ARD, X, theta, noise)
    anorm = np.linalg.norm(K, 1)
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    rcond, info = lapack.dpocon(L, anorm, uplo=b"L")
    assert info == 0
    alpha = sla.cho_solve((L, True), om, check_finite=False)
    kr, vr = O.query(O.SE_ARD, X, theta, L, alpha, Xq)
    mu, s2 = synth.finish_query(kr, vr, mean, noise)
    return O.log_lik(L, om, alpha), mu, s2, 1.0 / rcond


def test_gpu_c2_operating_points_vs_lapack(engine_lib):
    """config 2 "also at 16 random theta in U[-1,1]^7 to exercise conditioning", and (missing 1) at
    the theta a KernelLFOpt fit ends on — the point a fitted model lives at (kernel_lf_opt.hpp:60-69), far from theta = 0
    (log-lik -500 -> +3800).  N = 4096, D = 6, noise 0.01; the engine multiplies by explicit 64 x 64 block inverses (error
    ~ cond(L_bb) eps), so this is where the 1e-8 bar on mu / sigma^2 (gp.hpp:613-624) is at risk.  Per theta, against
    LAPACK on the host: mu (floor 1e-3) and sigma^2 (incl. + noise) on 256 points 1e-8, log-lik 1e-10; the gradient
    (gp.hpp:285-311) 1e-6 at four of the random theta and at the fitted one.  Worst errors and cond_1(K) are printed."""
    from limbo_amd import hpfit

    X, Y = synth.make_problem("c2")
    assert X.shape == (4096, 6)
    om, mean = synth.obs_mean_data(Y)
    noise = 0.01
    rng = np.random.default_rng(20260927)
    thetas = rng.uniform(-1.0, 1.0, size=(16, 7))
    Xq = rng.uniform(0, 1, size=(256, 6))
    # the fit of BASELINE configs[1] by the reference's protocol: Rprop 50 iterations x 10 restarts (lock-step)
    G = 10
    hs = []
    for _ in range(G):
        hh = _capi.Handle(engine_lib)
        hh.set_data(X, om)
        hs.append(hh)
    inits = np.random.default_rng(7).uniform(-1e-2, 1e-2, size=(G, 7))  # parallel_repeater.hpp:66,88 (epsilon 1e-2)
    inits[0] = 0.0
    bp, bl = hpfit.kernel_lf_opt_lockstep(hs, O.SE_ARD, inits, noise=noise, optimize_noise=False, iterations=50)
    th_fit = bp[int(np.argmax(bl))].copy()
    assert all(hh.flow_retries() == 0 for hh in hs)
    for hh in hs[1:]:
        hh.close()
    h = hs[0]
    worst = dict(mu=0.0, s2=0.0, ll=0.0, grad=0.0, cond=0.0)
    rows = []
    for i, th in enumerate(list(thetas) + [th_fit]):
        fitted = i == len(thetas)
        h.set_kernel(O.SE_ARD, th, noise)
        assert h.compute() == 0
        ll = h.log_lik()
        kta, var = h.query_batch(Xq)
        mu, s2 = synth.finish_query(kta, var, mean, noise)
        ll_r, mu_r, s2_r, cond = _lapack_posterior_se_ard(X, th, noise, om, mean, Xq)
        e = dict(mu=relerr(mu, mu_r, floor=1e-3), s2=relerr(s2, s2_r), ll=abs(ll - ll_r) / abs(ll_r), cond=cond, grad=0.0)
        if fitted or i < 4:
            g = h.log_lik_grad(False)
            _, g_r, _, _ = _lapack_grad_se_ard(X, th, noise, om, False)
            e["grad"] = relerr_norm(g, g_r)
        rows.append((("fit" if fitted else f"{i:3d}"), e, ll_r))
        for k in worst:
            worst[k] = max(worst[k], e[k])
    for tag, e, ll_r in rows:
        print(f"C2 theta {tag}: cond_1(K) {e['cond']:.2e}  log-lik {ll_r:10.2f}  mu {e['mu']:.2e}  sigma^2 {e['s2']:.2e}  "
              f"log-lik err {e['ll']:.2e}  grad {e['grad']:.2e}")
    print(f"C2 operating points, worst: mu {worst['mu']:.2e}  sigma^2 {worst['s2']:.2e}  log-lik {worst['ll']:.2e}  "
          f"grad {worst['grad']:.2e}  cond_1(K) up to {worst['cond']:.2e};  fitted theta {np.round(th_fit, 4).tolist()} log-lik {bl.max():.2f}")
    assert bl.max() > 0.0  # the fit moved far from theta = 0 (log-lik there is about -500)
    assert worst["mu"] < PC.TOL_MU and worst["s2"] < PC.TOL_VAR and worst["ll"] <= PC.TOL_LL and worst["grad"] < PC.TOL_GRAD, worst
    assert h.flow_retries() == 0
    h.close()


def test_gpu_c5_noise_1e_10_vs_mpmath_truth(engine_lib, oracle_lib):
    """VERDICT r4 (weak 1b): at bench.cpp:70's noise 1e-10 test_gpu_c5_add_sample_loop can hold the engine to the reference only
    at L 1e-6 / mu 1e-4 / sigma^2 1e-6 (cond(K) ~ 2e10).  This adjudicates with the truth: tests/golden/c5_truth_n200_noise1e-10.npz
    (60-digit mpmath from the same double inputs, oracle/make_golden_c5.py).  The same add_sample loop (10 samples, then 190
    appended: bench.cpp:66-67,83-84 / gp.hpp:126-152,573-603) on the engine, on the reference itself (oracle/_ref, where
    present) and on the C restatement; asserted: |engine - truth| <= 4 |reference - truth| in the max-norm for L, mu and
    sigma^2 (floors: 1e-13 of max|L|, 1e-12 of max|y|, 1e-14 absolute — never reached in practice), and the same for a full
    compute() of the 200 samples."""
    from tests.util import load, GOLDEN

    z = load(GOLDEN / "c5_truth_n200_noise1e-10.npz")
    X, Y, Xq, th, noise = z["X"], z["Y"], z["Xq"], z["theta"], float(z["noise"])
    n0, n1, D = 10, 200, 6
    # the fixture is the draw of test_gpu_c5_add_sample_loop
    rng = np.random.default_rng(2026)
    assert np.array_equal(X, rng.uniform(0, 1, size=(n1, D))) and np.array_equal(Y, synth.hartmann6(X)[:, None])
    om_all, mean = synth.obs_mean_data(Y)
    yscale = float(np.max(np.abs(Y)))

    def errors(L, mu, s2):
        return (float(np.max(np.abs(np.tril(L) - z["L"])) / np.max(np.abs(z["L"]))),
                float(np.max(np.abs(mu - z["mu"])) / yscale), float(np.max(np.abs(s2 - z["sigma2"]))))

    def loop(lib):
        om0, _ = synth.obs_mean_data(Y[:n0])
        g = new_gp(lib, O.SE_ARD, X[:n0], om0, th, noise)
        assert g.compute() == 0
        for n in range(n0, n1):
            om, _ = synth.obs_mean_data(Y[: n + 1])
            assert g.add_sample(X[n], om) == 0
        k, v = g.query_batch(Xq)
        out = errors(g.get_L(), *synth.finish_query(k, v, mean, noise))
        g.close()
        return out

    def full(lib):
        g = new_gp(lib, O.SE_ARD, X, om_all, th, noise)
        assert g.compute() == 0
        k, v = g.query_batch(Xq)
        out = errors(g.get_L(), *synth.finish_query(k, v, mean, noise))
        g.close()
        return out

    e_gpu, e_orc = loop(engine_lib), loop(oracle_lib)
    f_gpu, f_orc = full(engine_lib), full(oracle_lib)
    e_ref = f_ref = None
    if OB.ref_available():
        r = OB.RefGP(O.SE_ARD, D, 1, noise=noise)
        r.set_h_params(th)
        r.compute(X[:n0], Y[:n0])
        for n in range(n0, n1):
            r.add_sample(X[n], Y[n])
        e_ref = errors(r.matrixL(), *r.query(Xq))
        r2 = OB.RefGP(O.SE_ARD, D, 1, noise=noise)
        r2.set_h_params(th)
        r2.compute(X, Y)
        f_ref = errors(r2.matrixL(), *r2.query(Xq))
    fmt = lambda e: "L %.2e  mu %.2e  sigma^2 %.2e" % e  # noqa: E731
    print("C5 noise 1e-10, distance from the 60-digit truth (max-norm: L / max|L|, mu / max|y|, sigma^2 absolute)")
    print("  add_sample loop: engine", fmt(e_gpu), "| reference", fmt(e_ref) if e_ref else "-", "| C restatement", fmt(e_orc))
    print("  full compute   : engine", fmt(f_gpu), "| reference", fmt(f_ref) if f_ref else "-", "| C restatement", fmt(f_orc))
    floors = (1e-13, 1e-12, 1e-14)
    for got, refs in ((e_gpu, (e_ref, e_orc)), (f_gpu, (f_ref, f_orc))):
        yard = refs[0] if refs[0] is not None else refs[1]  # the reference itself where it travels, else its restatement
        for q in range(3):
            assert got[q] <= max(4.0 * yard[q], floors[q]), (q, got, yard)


_CHILD = r"""
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np
from limbo_amd import _capi, synth
import ctypes as C
eng = _capi.load_engine()
N, reps, start = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
X, Y = synth.make_problem("c4", N=N)
om, _ = synth.obs_mean_data(Y)
h = _capi.Handle(eng); h.set_data(X, om); h.set_kernel(synth.SE_ARD, np.zeros(7), 0.01)
while time.time() < start:
    pass
t0 = time.perf_counter()
lls, infos = [], []
for _ in range(reps):
    infos.append(h.compute()); lls.append(h.log_lik())
dt = time.perf_counter() - t0
k, v = h.query_batch(X[:5] + 0.01)  # (a one-launch sweep: a data-flow launch of its own)
w = C.c_int64(); eng.fn("xproc_waits")(C.byref(w))
print(json.dumps({"ll": [x.hex() for x in lls], "info": infos, "s": dt, "t_end": time.time(), "reruns": h.handover_reruns(),
                  "retries": h.flow_retries(), "xproc_waits": w.value, "var": [float(x).hex() for x in v]}))
"""


def test_gpu_two_processes_share_one_gpu():
    """VERDICT r5, missing 4: two BO PROCESSES on one MI355X — an ordinary way to run limbo experiments; the reference's
    gp.hpp:565 is re-entrant across processes.  The engine's data-flow launches wait for each other inside a launch and the gate
    that orders them is per process: two processes used to interleave their chains, run into the bounded polls and fall back
    to full re-runs.  Round 6: while another process has a handle on the GPU every data-flow launch scope runs under
    flock(/dev/shm/limbo_amd.gpu-<bus id>.lock), launch and host wait.  Two children x 50 evaluations at N = 2048 started at
    the same moment, NO switch set by hand: every log-lik bitwise the solo run's, no re-run, no sweep retry, the lock seen at
    work in both, >= 200 evaluations/s in all (solo: ~2400/s; taking turns costs a host wait per launch scope)."""
    import json
    import time

    env = dict(os.environ)
    for k in list(env):
        if k.startswith("GPE_"):
            del env[k]

    def spawn(start):
        return subprocess.Popen([sys.executable, "-c", _CHILD, str(ROOT), "2048", "50", repr(start)], stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, env=env, cwd=str(ROOT))

    solo = spawn(time.time())
    so, se = solo.communicate(timeout=600)
    assert solo.returncode == 0, se[-3000:]
    ref = json.loads(so.strip().splitlines()[-1])
    assert ref["reruns"] == 0 and ref["retries"] == 0 and all(i == 0 for i in ref["info"]) and len(set(ref["ll"])) == 1
    assert ref["xproc_waits"] == 0  # alone: nothing is ever locked
    start = time.time() + 6.0  # (both children import, upload and factor once before they start together)
    kids = [spawn(start), spawn(start)]
    outs = []
    for k in kids:
        o, e = k.communicate(timeout=900)
        assert k.returncode == 0, e[-3000:]
        outs.append((json.loads(o.strip().splitlines()[-1]), e))
    t_all = max(o["t_end"] for o, _ in outs) - start
    rate = 100.0 / t_all
    print(f"two processes on one GPU: {rate:.0f} evaluations/s in all (solo {50 / ref['s']:.0f}/s), scopes under the inter-process lock: "
          f"{[o['xproc_waits'] for o, _ in outs]}, re-runs {[o['reruns'] for o, _ in outs]}")
    for o, e in outs:
        assert o["ll"] == ref["ll"] and o["var"] == ref["var"], "results differ from the solo run"
        assert o["reruns"] == 0 and o["retries"] == 0 and all(i == 0 for i in o["info"]), (o["reruns"], o["retries"], e[-1000:])
        assert o["xproc_waits"] > 0 and "another process is using this GPU" in e
    assert rate >= 200.0, rate


@pytest.mark.parametrize("ranks", [2, 8])
def test_gpu_bench_ranks_on_one_gpu(ranks):
    """bench.py's world > 1 branch (one process per GPU under torch.distributed.run, barrier + max-over-ranks timing, the
    all-gather arg-max of tools/parallel.hpp:169-191) executed before the first 8-GPU run: 2 and 8 ranks — the driver's
    own command line for N = 8 — share the one visible GPU, the collectives travel over gloo on CPU tensors
    (--dist-backend gloo; the driver's runs use RCCL).  Eight PROCESSES on one GPU used to need the schedules without data-flow
    launches set by hand (rounds 4-5: GPE_TAIL_MAX=0 GPE_PANEL256=0 GPE_FLOW_SOLVE=0); since round 6 they take turns through the
    inter-process lock (include/gpe.h: gpe_xproc_waits) and run the production schedule — on the node every rank has its own GPU
    and nothing is locked; what is rehearsed here is the distributed plumbing: n_gpus, the 64 GPs of configs[3] dealt 8 per rank, the arg-max owner,
    and (round 6) configs[2]'s query points dealt over the ranks with every rank factoring its replica (`config3_sharded`)."""
    import json
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
           "--dist-backend", "gloo", "--c3-n", "2048", "--c3-m", "5003"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] * out["ms_per_step"] * 1e-3 - float(ranks)) < 1e-6  # ranks x K steps / max time
    assert out["config4"]["gps_total"] == 8 * ranks and out["config4"]["value"] > 0
    assert out["argmax"]["owner_rank"] in range(ranks) and np.isfinite(out["argmax"]["best_log_lik"])
    assert set(out["collectives"]["executed"]) >= {"barrier", "all_gather", "all_reduce"} and out["collectives"]["world"] == ranks
    assert "roofline" in out and "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert out["handover_reruns"] == 0 and out["flow_retries"] == 0  # (the ranks took turns instead of starving each other)
    # round 6: configs[2] sharded over the query points (a replica per rank, row_slice of the points, one all-gather) at a reduced N
    c3 = out["config3_sharded"]
    assert c3["points"] == 5003 and c3["points_per_rank"] == 5003 // ranks + (1 if 5003 % ranks else 0) and c3["value"] > 0
    assert c3["gathered_equals_own_answer_bitwise"] is True and np.isfinite(c3["log_lik"]) and c3["gather_s_max_over_ranks"] >= 0
    print(f"{ranks} ranks on one GPU: {out['value']:.1f} evaluations/s in all, config4 {out['config4']['value']:.0f}/s, arg-max owner rank {out['argmax']['owner_rank']}")


def test_gpu_bench_one_rank_rccl():
    """The RCCL branch of bench.py executed before the driver's 8-GPU run does it for the first time: `--force-dist` initialises
    the `nccl` (= RCCL) process group even with one rank, so barrier / all_gather / all_reduce run on cuda:0 tensors exactly
    as they will at N > 1 (tools/parallel.hpp:169-191: the arg-max over the restarts).  One JSON line, the collectives
    listed as executed, the value of the same order as a run without the group (the collectives are microseconds)."""
    import json
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    common = ["--gpus", "1", "--steps", "20", "--warmup", "3", "--no-extras", "--no-cpu-baseline", "--no-roofline"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--force-dist", "--c3-n", "2048", "--c3-m", "3000"] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["collectives"]["backend"] == "nccl" and out["collectives"]["tensors_on"].startswith("cuda")
    assert {"barrier", "all_gather", "all_reduce"} <= set(out["collectives"]["executed"])
    assert out["argmax"]["owner_rank"] == 0 and np.isfinite(out["argmax"]["best_log_lik"])
    assert out["config4"]["gps_total"] == 8 and out["config4"]["value"] > 0
    assert out["config3_sharded"]["points_per_rank"] == 3000 and out["config3_sharded"]["gathered_equals_own_answer_bitwise"] is True
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    plain = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    print(f"one rank with the RCCL group: {out['value']:.1f} evaluations/s, without: {plain['value']:.1f}")
    # (a sanity bound, not a measurement: 20 steps of 1 ms; before the collectives were warmed up outside the timed region the first
    # all_gather's set-up alone cost 10 % of it — profiles/r06_bench_ranks.log)
    assert plain["collectives"]["executed"] == [] and abs(out["value"] - plain["value"]) <= 0.20 * plain["value"]
    assert out["log_lik"] == plain["log_lik"]


@pytest.mark.parametrize("N,G,kind,on,P", [(2048, 8, O.SE_ARD, False, 1), (700, 5, O.MATERN52, True, 2), (1100, 18, O.SE_ARD, True, 1),
                                            (300, 3, O.SE_ARD, False, 1),
                                            # round 6: a batch of fewer than four members CUTS k ranges — every member counts its chunks in
                                            # counter words of its own (inv2.hip), ragged order
                                            (2100, 3, O.SE_ARD, True, 1)])
def test_gpu_batch_hp_objective_vs_single_and_oracle(engine_lib, oracle_lib, N, G, kind, on, P):
    """gpe_batch_hp_objective — G restarts of KernelLFOptimization::operator() (kernel_lf_opt.hpp:77-92,
    parallel_repeater.hpp:84-105) stepped by one launch sequence: every member's (log-lik, gradient) equals the
    single-handle gpe_hp_objective to rounding and the oracle's to the usual tolerances; K^-1 of a member equals the oracle's;
    the batch is bitwise reproducible; a second, different batch on the same handles works (resident buffers)."""
    rng = np.random.default_rng(N + G)
    D = 6 if kind == O.SE_ARD else 3
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) + 0.05 * rng.normal(size=N) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Y)
    nt = D + 1 if kind == O.SE_ARD else 2
    hs = []
    for g in range(G):
        h = _capi.Handle(engine_lib)
        h.set_data(X, om * (1.0 + 0.1 * g))
        hs.append(h)
    single = _capi.Handle(engine_lib)
    orc = _capi.Handle(oracle_lib)
    for rnd in range(2):
        th = rng.uniform(-0.4, 0.3, size=(G, nt))
        nz = 0.01 * np.exp(rng.uniform(-0.5, 0.5, size=G))
        lik, grad, st = _capi.batch_hp_objective(hs, kind, th, nz, optimize_noise=on, want_grad=True)
        lik2, grad2, st2 = _capi.batch_hp_objective(hs, kind, th, nz, optimize_noise=on, want_grad=True)
        assert all(s == 0 for s in st) and np.array_equal(lik, lik2) and np.array_equal(grad, grad2)
        for g in sorted(set([0, G // 2, G - 1])):
            single.set_data(X, om * (1.0 + 0.1 * g))
            l1, g1, info = single.hp_objective(kind, th[g], nz[g], optimize_noise=on, want_grad=True)
            assert info == 0 and abs(lik[g] - l1) <= 1e-11 * abs(l1) and relerr_norm(grad[g], g1) < 1e-9
            if N <= 1100:
                orc.set_data(X, om * (1.0 + 0.1 * g))
                lo, go, _ = orc.hp_objective(kind, th[g], nz[g], optimize_noise=on, want_grad=True)
                assert abs(lik[g] - lo) <= PC.TOL_LL * abs(lo) and relerr_norm(grad[g], go) < PC.TOL_GRAD
                assert relerr_norm(hs[g].get_Kinv(), orc.get_Kinv()) < 1e-8
        # the value-only form
        lik3, none, st3 = _capi.batch_hp_objective(hs, kind, th, nz, optimize_noise=on, want_grad=False)
        assert none is None and np.array_equal(lik3, lik)
    assert all(h.flow_retries() == 0 for h in hs)
    for h in hs + [single, orc]:
        h.close()
