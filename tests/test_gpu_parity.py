"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, through the C-ABI, against
(1) the committed mpmath / LAPACK goldens, (2) the CPU oracle on the same seeded inputs at
sizes the oracle finishes in seconds, (3) size-independent properties at BASELINE sizes.
Tolerances: mu, sigma^2 <= 1e-8 rel; log-lik <= 1e-10 rel; grad <= 1e-6 rel."""
from pathlib import Path

import numpy as np
import pytest

from limbo_amd import _capi
from oracle import binding as OB
from oracle import np_oracle as O
from tests import parity_checks as PC
from tests.util import golden_files, new_gp, relerr, relerr_norm

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("path", golden_files("mp_"), ids=lambda p: p.stem)
def test_gpu_vs_mpmath_golden(engine_lib, path):
    PC.check_against_mp_golden(engine_lib, path)


@pytest.mark.parametrize("path", golden_files("np_"), ids=lambda p: p.stem)
def test_gpu_vs_lapack_golden(engine_lib, path):
    PC.check_against_np_golden(engine_lib, path)


@pytest.mark.parametrize("kind", [O.SE_ARD, O.MATERN52, O.MATERN32, O.EXP])
def test_gpu_kernel_matrix_vs_oracle(engine_lib, oracle_lib, kind):
    """kernel build (gp.hpp:556-562): elementwise vs the oracle, ragged N, exact symmetry."""
    rng = np.random.default_rng(200 + kind)
    N, D = 333, 6
    X = rng.uniform(-2, 2, size=(N, D))
    nt = D + 1 if kind == O.SE_ARD else 2
    th = rng.uniform(-1, 1, size=nt)
    g = new_gp(engine_lib, kind, X, np.zeros((N, 1)), th, 0.03)
    o = new_gp(oracle_lib, kind, X, np.zeros((N, 1)), th, 0.03)
    Kg, Ko = g.get_K(), o.get_K()
    assert relerr(Kg, Ko, floor=1e-30) < 5e-14
    assert np.array_equal(Kg, Kg.T)
    g.close()
    o.close()


@pytest.mark.parametrize("N,D,P,kind", [(1, 2, 1, O.SE_ARD), (63, 3, 1, O.SE_ARD), (64, 3, 2, O.MATERN52),
                                        (65, 1, 1, O.SE_ARD), (257, 6, 3, O.SE_ARD), (700, 12, 1, O.MATERN52),
                                        (1100, 6, 2, O.SE_ARD)])
def test_gpu_vs_oracle_full_path(engine_lib, oracle_lib, N, D, P, kind):
    """compute / log-lik / grad / query against the oracle on identical inputs, including
    block-edge sizes (63, 64, 65, 257: the 64-column and 256-column panel boundaries)."""
    rng = np.random.default_rng(N * 7 + D)
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) + 0.05 * rng.normal(size=N) for p in range(P)], axis=1)
    om, mean = O.obs_mean_data(Y)
    nt = D + 1 if kind == O.SE_ARD else 2
    th = rng.uniform(-0.5, 0.3, size=nt)
    noise = 0.01
    g = new_gp(engine_lib, kind, X, om, th, noise)
    o = new_gp(oracle_lib, kind, X, om, th, noise)
    assert g.compute() == 0 and o.compute() == 0
    Lg, Lo = g.get_L(), o.get_L()
    assert np.max(np.abs(Lg - Lo)) < 1e-10 * np.max(np.abs(Lo))
    assert np.all(np.triu(Lg, 1) == 0.0)
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    llg, llo = g.log_lik(), o.log_lik()
    assert abs(llg - llo) <= PC.TOL_LL * max(1.0, abs(llo))
    gg, go = g.log_lik_grad(True), o.log_lik_grad(True)
    assert relerr_norm(gg, go) < PC.TOL_GRAD
    Xq = rng.uniform(0, 1, size=(130, D))
    Xq[:3] = X[:3] if N >= 3 else Xq[:3]
    kg, vg = g.query_batch(Xq)
    ko, vo = o.query_batch(Xq)
    mug, s2g = O.finish_query(kg, vg, mean, noise)
    muo, s2o = O.finish_query(ko, vo, mean, noise)
    assert relerr(mug, muo, floor=1e-3) < PC.TOL_MU
    assert relerr(s2g, s2o) < PC.TOL_VAR
    assert relerr_norm(g.get_Kinv(), o.get_Kinv()) < 1e-8
    g.close()
    o.close()


@pytest.mark.parametrize("N,D,P,kind,on", [(40, 4, 2, O.SE_ARD, True), (130, 3, 1, O.MATERN52, False),
                                           (257, 6, 3, O.SE_ARD, True), (333, 2, 1, O.EXP, True),
                                           (200, 5, 8, O.MATERN32, False), (150, 3, 11, O.SE_ARD, True)])
def test_gpu_loo_cv_vs_oracle(engine_lib, oracle_lib, N, D, P, kind, on):
    """compute_log_loo_cv / compute_kernel_grad_log_loo_cv (gp.hpp:339-402): the device's one-product
    form against the oracle's literal per-parameter Zeta products."""
    rng = np.random.default_rng(N + 31 * D)
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) + 0.05 * rng.normal(size=N) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Y)
    nt = D + 1 if kind == O.SE_ARD else 2
    th = rng.uniform(-0.5, 0.3, size=nt)
    g = new_gp(engine_lib, kind, X, om, th, 0.02)
    o = new_gp(oracle_lib, kind, X, om, th, 0.02)
    assert g.compute() == 0 and o.compute() == 0
    lg, lo = g.log_loo_cv(), o.log_loo_cv()
    assert abs(lg - lo) <= 1e-9 * max(1.0, abs(lo))
    gg, go = g.log_loo_cv_grad(on), o.log_loo_cv_grad(on)
    assert relerr_norm(gg, go) < PC.TOL_GRAD
    assert relerr_norm(g.get_loo_weights(), o.get_loo_weights()) < 1e-8
    # the log-likelihood gradient still works afterwards (the LOO path reuses the L^-1 scratch)
    assert relerr_norm(g.log_lik_grad(on), o.log_lik_grad(on)) < PC.TOL_GRAD
    assert relerr_norm(g.get_Kinv(), o.get_Kinv()) < 1e-8
    g.close()
    o.close()


@pytest.mark.parametrize("kind,on", [(O.SE_ARD, True), (O.MATERN52, False)])
def test_gpu_loo_grad_fd(engine_lib, kind, on):
    PC.check_loo_grad_fd(engine_lib, kind, on)


@pytest.mark.parametrize("dup", [False, True])
def test_gpu_incremental_vs_full(engine_lib, dup):
    PC.check_incremental_vs_full(engine_lib, dup=dup)


def test_gpu_incremental_across_capacity_growth(engine_lib):
    """add_sample past the initial capacity (reallocation keeps X and L)."""
    PC.check_incremental_vs_full(engine_lib, kind=O.MATERN52, n0=250, n1=270, D=2, P=1, seed=17)


def test_gpu_add_sample_from_empty(engine_lib):
    PC.check_add_sample_from_empty(engine_lib)


@pytest.mark.parametrize("kind,on", [(O.SE_ARD, False), (O.SE_ARD, True), (O.MATERN52, True)])
def test_gpu_grad_fd(engine_lib, kind, on):
    PC.check_grad_fd(engine_lib, kind, on)


def test_gpu_update_alpha_and_clone(engine_lib):
    PC.check_update_alpha_and_clone(engine_lib)


def test_gpu_host_K(engine_lib):
    PC.check_host_K(engine_lib)


def test_gpu_not_pd(engine_lib):
    PC.check_not_pd(engine_lib)


def test_gpu_batch_compute_matches_single(engine_lib):
    """independent GPs (multi_gp.hpp:124-126) enqueued together == one at a time."""
    from limbo_amd import _capi

    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, size=(300, 6))
    hs, ref = [], []
    for gidx in range(6):
        Y = rng.normal(size=(300, 1))
        om, _ = O.obs_mean_data(Y)
        th = rng.uniform(-0.3, 0.3, size=7)
        hs.append(new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01))
        r = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
        r.compute()
        ref.append(r.log_lik())
        r.close()
    st = _capi.batch_compute(hs)
    assert st == [0] * 6
    ll = _capi.batch_log_lik(hs)
    # one launch sequence steps all six (gridDim.z = GP): same arithmetic per GP, but the one-stream schedule and the
    # tile shapes chosen for a six-fold launch differ from a lone evaluation's => equal to rounding, not bitwise
    assert np.max(np.abs(ll - np.array(ref)) / np.abs(np.array(ref))) < 1e-12
    assert _capi.batch_compute(hs) == [0] * 6
    assert np.array_equal(_capi.batch_log_lik(hs), ll)  # run to run: bitwise
    for h in hs:
        h.close()


def test_gpu_deterministic(engine_lib):
    """two runs on the same inputs are bitwise identical (fixed-order reductions, no atomics)."""
    X, Y = O.make_problem("c2", N=900)
    om, _ = O.obs_mean_data(Y)
    out = []
    for _ in range(2):
        h = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(7), 0.01)
        h.compute()
        out.append((h.log_lik(), h.log_lik_grad(True), h.get_alpha()))
        h.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


def test_gpu_c2_full_size_properties(engine_lib):
    """BASELINE config C2 (N=4096, D=6, SE-ARD, fp64) — size-independent properties:
    L L^T reproduces K, K alpha = obs_mean, K^-1 K = I on sampled columns, and the
    log-likelihood matches LAPACK (scipy) to 1e-10."""
    import scipy.linalg as sla

    X, Y = O.make_problem("c2")
    om, mean = O.obs_mean_data(Y)
    th = np.zeros(7)
    h = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
    assert h.compute() == 0
    L = h.get_L()
    K = O.kernel_matrix(O.SE_ARD, X, th, 0.01)
    rng = np.random.default_rng(0)
    cols = rng.integers(0, 4096, size=32)
    LLt = L @ L[cols].T
    assert np.max(np.abs(LLt - K[:, cols])) < 1e-12 * np.max(np.abs(K))
    a = h.get_alpha()
    assert np.linalg.norm(K @ a - om) < 1e-9 * np.linalg.norm(om)
    Ln = sla.cholesky(K, lower=True)
    ll_ref = O.log_lik(Ln, om, sla.cho_solve((Ln, True), om))
    assert abs(h.log_lik() - ll_ref) <= 1e-10 * abs(ll_ref)
    # mu / sigma^2 on 256 test points vs LAPACK
    Xq = rng.uniform(0, 1, size=(256, 6))
    kta, var = h.query_batch(Xq)
    kr, vr = O.query(O.SE_ARD, X, th, Ln, sla.cho_solve((Ln, True), om), Xq)
    mu, s2 = O.finish_query(kta, var, mean, 0.01)
    mur, s2r = O.finish_query(kr, vr, mean, 0.01)
    assert relerr(mu, mur, floor=1e-3) < PC.TOL_MU
    assert relerr(s2, s2r) < PC.TOL_VAR
    h.close()


def test_gpu_c3_full_size_properties(engine_lib):
    """BASELINE config C3 (N=16384, D=12, Matern-5/2, fp64; 4096 of the 100k queries here) —
    size-independent properties: K alpha = obs_mean on sampled rows, at training points
    mu ~ y and sigma^2 <= 2 (noise + 1e-8) (src/tests/test_gp.cpp:448-511), batched == chunked."""
    rng = np.random.default_rng(3)
    N, D = 16384, 12
    X = rng.uniform(0, 1, size=(N, D))
    Y = (np.cos(2 * X).sum(axis=1) + 0.05 * rng.normal(size=N))[:, None]
    om, mean = O.obs_mean_data(Y)
    th = np.zeros(2)
    noise = 0.01
    h = new_gp(engine_lib, O.MATERN52, X, om, th, noise)
    assert h.compute() == 0
    a = h.get_alpha()
    rows = rng.integers(0, N, size=48)
    # K[rows, :] on the host (Matern 5/2, l = 1, sigma_f = 1), + (noise + 1e-8) on the diagonal
    d = np.sqrt(((X[rows, None, :] - X[None, :, :]) ** 2).sum(-1))
    t1 = np.sqrt(5.0) * d
    Kr = (1 + t1 + 5.0 * d * d / 3.0) * np.exp(-t1)
    Kr[np.arange(len(rows)), rows] += noise + 1e-8
    assert np.max(np.abs(Kr @ a[:, 0] - om[rows, 0])) < 1e-8 * np.max(np.abs(om))
    assert np.isfinite(h.log_lik())
    kta, var = h.query_batch(X[rows])
    mu, s2 = O.finish_query(kta, var, mean, noise)
    assert np.max(np.abs(mu[:, 0] - Y[rows, 0])) < 1.0
    assert np.all(s2 <= 2.0 * (noise + 1e-8))
    Xq = rng.uniform(0, 1, size=(4096, D))
    k1, v1 = h.query_batch(Xq)
    k2, v2 = h.query_batch(Xq[:1000])
    k3, v3 = h.query_batch(Xq)
    assert np.array_equal(k1, k3) and np.array_equal(v1, v3)  # bitwise reproducible from call to call
    # a different batch size may pick other matrix-core tile shapes for the solve (another summation order): equal to rounding
    assert np.array_equal(k1[:1000], k2) and relerr(v1[:1000], v2) < 1e-10
    assert np.all(np.isfinite(v1)) and np.all(v1 > -1e-9)
    h.close()


@pytest.mark.parametrize("N,D,mp", [(300, 2, 100), (257, 6, 64), (500, 12, 200), (90, 3, 90), (1000, 3, 150)])
def test_gpu_sparsify_vs_oracle(engine_lib, oracle_lib, N, D, mp):
    """SparsifiedGP::_sparsify (sparsified_gp.hpp:157-183): the device's cached-density form keeps
    exactly the samples the reference's rescan-everything loop keeps."""
    from limbo_amd import _capi
    rng = np.random.default_rng(7 * N + D)
    X = rng.uniform(0, 1, size=(N, D))
    X[: N // 4] = 0.3 + 0.02 * rng.normal(size=(N // 4, D))
    kg = _capi.sparsify(engine_lib, X, mp)
    ko = _capi.sparsify(oracle_lib, X, mp)
    assert np.array_equal(kg, ko)


def test_gpu_sparsify_full_size(engine_lib):
    """N = 4096 -> 200 (the reference's default max_points): result size, ordering, and the thinning
    property — the minimum nearest-neighbour distance does not decrease."""
    from limbo_amd import _capi
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, size=(4096, 6))
    keep = _capi.sparsify(engine_lib, X, 200)
    assert len(keep) == 200 and np.all(np.diff(keep) > 0)

    def min_nn(Z):
        d = np.sqrt(((Z[:, None, :] - Z[None, :, :]) ** 2).sum(axis=2))
        np.fill_diagonal(d, np.inf)
        return d.min()

    assert min_nn(X[keep]) > min_nn(X[:1500])


def test_gpu_vs_oracle_n2048_regular_panels(engine_lib, oracle_lib):
    """N = 2048 = 8 regular outer panels: every panel goes through the look-ahead schedule, the fused
    next-panel update (k_upd_fused) and the data-flow backward sweep — factor, alpha, log-lik and queries
    against the oracle on the same inputs."""
    rng = np.random.default_rng(2048)
    N, D = 2048, 6
    X = rng.uniform(0, 1, size=(N, D))
    Y = (np.cos(3 * X.sum(axis=1)) + 0.05 * rng.normal(size=N))[:, None]
    om, mean = O.obs_mean_data(Y)
    th = rng.uniform(-0.4, 0.2, size=D + 1)
    g = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
    o = new_gp(oracle_lib, O.SE_ARD, X, om, th, 0.01)
    assert g.compute() == 0 and o.compute() == 0
    Lg, Lo = g.get_L(), o.get_L()
    assert np.max(np.abs(Lg - Lo)) < 1e-10 * np.max(np.abs(Lo))
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    llg, llo = g.log_lik(), o.log_lik()
    assert abs(llg - llo) <= PC.TOL_LL * max(1.0, abs(llo))
    Xq = rng.uniform(0, 1, size=(200, D))
    kg, vg = g.query_batch(Xq)
    ko, vo = o.query_batch(Xq)
    mug, s2g = O.finish_query(kg, vg, mean, 0.01)
    muo, s2o = O.finish_query(ko, vo, mean, 0.01)
    assert relerr(mug, muo, floor=1e-3) < PC.TOL_MU
    assert relerr(s2g, s2o) < PC.TOL_VAR
    # bitwise reproducible from run to run (fixed-order reductions, no atomics in any sum)
    assert g.compute() == 0
    assert np.array_equal(g.get_L(), Lg) and g.log_lik() == llg
    g.close()
    o.close()


@pytest.mark.parametrize("N,D,k,P", [(70, 3, 1, 1), (130, 1, 1, 1), (257, 6, 2, 2), (300, 12, 3, 1), (600, 6, 5, 1)])
def test_gpu_se_ard_lambda_vs_oracle(engine_lib, oracle_lib, N, D, k, P):
    """SquaredExpARD with k columns of Lambda (squared_exp_ard.hpp:109-126, :142-146): the device evaluates
    d^T (Lambda Lambda^T + diag(ell^-2)) d as a sum of squares over D + k rows (inputs + projections);
    the oracle forms M literally.  Every entry point the kernel reaches: K, L, alpha, log-lik, both
    gradients (one pair-sum pass per Lambda column), queries, add_sample, clone."""
    rng = np.random.default_rng(N * 3 + D + k)
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) + 0.05 * rng.normal(size=N) for p in range(P)], axis=1)
    om, mean = O.obs_mean_data(Y)
    th = np.concatenate([rng.uniform(-0.5, 0.3, size=D), rng.uniform(-1.0, 1.0, size=D * k), [rng.uniform(-0.2, 0.2)]])
    noise = 0.02
    g = new_gp(engine_lib, O.SE_ARD, X, om, th, noise)
    o = new_gp(oracle_lib, O.SE_ARD, X, om, th, noise)
    assert relerr(g.get_K(), o.get_K(), floor=1e-30) < 1e-11
    assert g.compute() == 0 and o.compute() == 0
    Lg, Lo = g.get_L(), o.get_L()
    assert np.max(np.abs(Lg - Lo)) < 1e-10 * np.max(np.abs(Lo))
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    llg, llo = g.log_lik(), o.log_lik()
    assert abs(llg - llo) <= PC.TOL_LL * max(1.0, abs(llo))
    for on in (False, True):
        gg, go = g.log_lik_grad(on), o.log_lik_grad(on)
        assert gg.size == th.size + on
        assert relerr_norm(gg, go) < PC.TOL_GRAD
        # per block of the parameter vector, so that a small block cannot hide behind a large one
        for sl in (slice(0, D), slice(D, D + D * k), slice(D + D * k, None)):
            assert relerr_norm(gg[sl], go[sl]) < 10 * PC.TOL_GRAD
    assert abs(g.log_loo_cv() - o.log_loo_cv()) <= 1e-9 * max(1.0, abs(o.log_loo_cv()))
    if N <= 300:  # the oracle's LOO gradient is T dense N^3 products
        assert relerr_norm(g.log_loo_cv_grad(True), o.log_loo_cv_grad(True)) < PC.TOL_GRAD
    Xq = rng.uniform(0, 1, size=(130, D))
    Xq[:3] = X[:3]
    kg, vg = g.query_batch(Xq)
    ko, vo = o.query_batch(Xq)
    mug, s2g = O.finish_query(kg, vg, mean, noise)
    muo, s2o = O.finish_query(ko, vo, mean, noise)
    assert relerr(mug, muo, floor=1e-3) < PC.TOL_MU
    assert relerr(s2g, s2o) < PC.TOL_VAR
    # incremental update with the same kernel (gp.hpp:573-603), then a clone answering the same queries
    xn = rng.uniform(0, 1, size=D)
    Y2 = np.vstack([Y, np.cos((np.arange(P) + 1) * xn.sum())[None, :]])
    om2, mean2 = O.obs_mean_data(Y2)
    assert g.add_sample(xn, om2) == 0 and o.add_sample(xn, om2) == 0
    assert np.max(np.abs(g.get_L() - o.get_L())) < 1e-9 * np.max(np.abs(Lo))
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    c = g.clone()
    kc, vc = c.query_batch(Xq)
    k2, v2 = g.query_batch(Xq)
    assert np.array_equal(kc, k2) and np.array_equal(vc, v2)
    c.close()
    g.close()
    o.close()


@pytest.mark.parametrize("on", [False, True])
def test_gpu_se_ard_lambda_grad_fd(engine_lib, on):
    """test_gp.cpp:131-271's finite-difference check with a D x 1 Lambda in the parameter vector."""
    PC.check_grad_fd(engine_lib, O.SE_ARD, on, lam=1)


def test_gpu_se_ard_parameter_count(engine_lib):
    """D + D k + 1 parameters, k = 0 .. D (squared_exp_ard.hpp:94); anything else is an argument error."""
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, size=(20, 3))
    om = rng.normal(size=(20, 1))
    for nt, ok in [(4, True), (7, True), (13, True), (5, False), (3, False), (16, False)]:
        g = new_gp(engine_lib, O.SE_ARD, X, om, np.zeros(nt), 0.01)
        try:
            rc = g.compute()
        except _capi.EngineError:
            rc = -1
        assert (rc == 0) == ok, (nt, rc)
        g.close()


@pytest.mark.parametrize("N,P", [(1, 1), (64, 2), (130, 11), (1100, 2), (2049, 1)])
def test_gpu_update_alpha_vs_oracle(engine_lib, oracle_lib, N, P):
    """recompute(true, false) (gp.hpp:241-252, :605-611) through the one-launch forward and backward sweeps:
    alpha and the log-likelihood against the oracle's substitution on the same factor; ragged last block,
    more outputs than one pass carries (P = 11 > 8: the per-block partial sums accumulate over passes)."""
    rng = np.random.default_rng(N + P)
    D = 3
    X = rng.uniform(0, 1, size=(N, D))
    Y1 = rng.normal(size=(N, P))
    Y2 = np.stack([np.sin((p + 2) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.1 * rng.normal(size=(N, P))
    om1, _ = O.obs_mean_data(Y1)
    om2, _ = O.obs_mean_data(Y2)
    th = np.array([-0.3, 0.1, -0.1, 0.05])
    g = new_gp(engine_lib, O.SE_ARD, X, om1, th, 0.02)
    o = new_gp(oracle_lib, O.SE_ARD, X, om1, th, 0.02)
    assert g.compute() == 0 and o.compute() == 0
    g.update_alpha(om2)
    o.update_alpha(om2)
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    assert abs(g.log_lik() - o.log_lik()) <= PC.TOL_LL * max(1.0, abs(o.log_lik()))
    g.close()
    o.close()


@pytest.mark.parametrize("N", [1, 63, 65, 700, 2049])
def test_gpu_point_queries_vs_batch_and_oracle(engine_lib, oracle_lib, N):
    """mu / sigma^2 for 1..8 points (N <= 256: the one-workgroup small path, csrc/small.hip; above: one-launch forward
    sweep, k_trsv_fwd_flow) against the oracle and against the same points inside a batch of 100 (blocked matrix
    solve): mu identical above 256 samples and equal to rounding below, sigma^2 to rounding."""
    rng = np.random.default_rng(N)
    D, P = 4, 2
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.05 * rng.normal(size=(N, P))
    om, mean = O.obs_mean_data(Y)
    th = rng.uniform(-0.4, 0.2, size=D + 1)
    g = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
    o = new_gp(oracle_lib, O.SE_ARD, X, om, th, 0.01)
    assert g.compute() == 0 and o.compute() == 0
    Xq = rng.uniform(0, 1, size=(100, D))
    Xq[0] = X[0]
    kb, vb = g.query_batch(Xq)
    ko, vo = o.query_batch(Xq)
    _, s2o = O.finish_query(ko, vo, mean, 0.01)
    for m in (1, 2, 5, 8):
        k1, v1 = g.query_batch(Xq[:m])
        # (a batch sums k*^T alpha over the points-contiguous layout, the handful over the sample-contiguous one, the
        #  one-launch small path in a third fixed order: equal to rounding, each bitwise reproducible)
        assert np.max(np.abs(k1 - kb[:m])) <= 1e-12 * max(np.max(np.abs(kb[:m])), 1.0)
        _, s2 = O.finish_query(k1, v1, mean, 0.01)
        _, s2b = O.finish_query(kb[:m], vb[:m], mean, 0.01)
        assert relerr(s2, s2b) < 1e-11
        assert relerr(s2, s2o[:m]) < PC.TOL_VAR
    # run-to-run bitwise reproducible
    k1, v1 = g.query_batch(Xq[:3])
    k2, v2 = g.query_batch(Xq[:3])
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
    g.close()
    o.close()


def _small_parity(engine_lib, oracle_lib, N=300, D=3, P=2, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, size=(N, D))
    Y = np.stack([np.cos((p + 1) * X.sum(axis=1)) for p in range(P)], axis=1) + 0.05 * rng.normal(size=(N, P))
    om, mean = O.obs_mean_data(Y)
    th = rng.uniform(-0.4, 0.2, size=D + 1)
    g = new_gp(engine_lib, O.SE_ARD, X, om, th, 0.01)
    o = new_gp(oracle_lib, O.SE_ARD, X, om, th, 0.01)
    assert g.compute() == 0 and o.compute() == 0
    assert np.max(np.abs(g.get_L() - o.get_L())) < 1e-10 * np.max(np.abs(o.get_L()))
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    assert abs(g.log_lik() - o.log_lik()) <= PC.TOL_LL * max(1.0, abs(o.log_lik()))
    assert relerr_norm(g.log_lik_grad(True), o.log_lik_grad(True)) < PC.TOL_GRAD
    assert relerr_norm(g.log_loo_cv_grad(True), o.log_loo_cv_grad(True)) < PC.TOL_GRAD
    assert relerr_norm(g.get_Kinv(), o.get_Kinv()) < 1e-8
    Xq = rng.uniform(0, 1, size=(5, D))
    for m in (1, 5):
        kg, vg = g.query_batch(Xq[:m])
        ko, vo = o.query_batch(Xq[:m])
        assert relerr(kg, ko, floor=1e-6) < PC.TOL_MU and relerr(vg + 0.01, vo + 0.01) < PC.TOL_VAR
    g.update_alpha(om[::-1].copy())
    o.update_alpha(om[::-1].copy())
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    xn = rng.uniform(0, 1, size=D)
    om2 = np.vstack([om[::-1], np.zeros((1, P))])
    assert g.add_sample(xn, om2) == 0 and o.add_sample(xn, om2) == 0
    assert relerr_norm(g.get_alpha(), o.get_alpha()) < 1e-7
    g.close()
    o.close()


@pytest.mark.parametrize("env", [{"GPE_FLOW_SOLVE": "0"}, {"GPE_LOOKAHEAD": "0"}, {"GPE_FUSE_PANEL": "0"},
                                 {"GPE_FUSE_DIAG": "0", "GPE_STOP_EVENT": "0"}, {"GPE_NBO": "192"},
                                 {"GPE_PANEL256": "0"}, {"GPE_TAIL_MAX": "0"}, {"GPE_TALL": "0"}, {"GPE_TAIL_LAG": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_gpu_alternate_schedules(engine_lib, oracle_lib, monkeypatch, env):
    """The switches read when a handle is created select the schedules that also serve as fall-backs (per-block
    sweeps beyond 256 blocks, one stream, unfused panel steps, a panel width the block-inverse path does not
    take): each must give the oracle's results too.  N = 520: three outer panels, ragged last block."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _small_parity(engine_lib, oracle_lib, N=520, seed=len(env))


_SWITCH_MEMO = []


def _switch_memo_dir(tmp_path_factory):
    if not _SWITCH_MEMO:
        _SWITCH_MEMO.append(tmp_path_factory.mktemp("switches"))
    return _SWITCH_MEMO[0]


@pytest.mark.parametrize("env", [{"GPE_QUERY_SWEEP": "0", "GPE_INV_PANELS": "0"}, {"GPE_QUERY_T": "0", "GPE_INV_OVERLAP": "0"},
                                 {"GPE_RHS_FMA": "2"}, {"GPE_RHS_FMA": "0"}, {"GPE_PANEL256": "0", "GPE_TAIL_MAX": "0"}, {"GPE_TAIL_MAX": "0"},
                                 {"GPE_EARLY_BULK_TILES": "0", "GPE_ROWS_TAIL": "0"},
                                 {"GPE_INV2": "0"}, {"GPE_HP_FUSED": "0", "GPE_INV2_MIN_N": "256"},
                                 {"GPE_FLOW_PARTITIONS": "0", "GPE_FLOW_XCD": "0", "GPE_FLOW_GATE": "0"},
                                 {"GPE_STREAM_PRIO": "0", "GPE_TAIL_GEN": "0", "GPE_TRACE": "1", "GPE_ROCTX": "1"},
                                 {"GPE_BATCH_SPLIT": "0", "GPE_BATCH_TAIL_TILES": "0", "GPE_BATCH_TAIL_MAX": "512"}, {"GPE_BATCH": "0"},
                                 {"GPE_SWEEP_M": "0", "GPE_XPROC_LOCK": "0", "GPE_TRI_MAP": "0", "GPE_RAGGED_FINISH": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_gpu_process_wide_switches(env, tmp_path_factory):
    """Switches that are read once per process (the blocked point-query solve, the in-panel substitution chain for K^-1; round 3:
    the sample-contiguous batched queries, K^-1's product after its chain, right-hand-side rows always / never as FMAs,
    the look-ahead stream released by every panel and obs_mean's rows by a launch of their own; round 4: 256-column panels to
    the end — one-launch (GPE_TAIL_MAX=0) or step by step (+ GPE_PANEL256=0): the schedules the data-flow launches replaced and
    what a hand-over timeout falls back to; round 5: K^-1 in its panel form instead of the recursion, the objective as
    separate calls instead of one enqueue, no CU-masked chain partitions / XCD-local sweeps / gate, default stream priorities
    with K built in front of the one-launch factorisation and the launch trace on, batched sequences unsplit / with the
    data-flow launches at any size / member by member): the same parity checks in a child, at a size with seven outer
    panels (N = 1700: ragged last panel), for the hyper-parameter objective through gpe_hp_objective (N = 1792: seven panels,
    the recursion's tree is unbalanced), and a batched factorisation (3 x N = 700) and objective (2 x N = 1024) — every
    switch of SWITCHES.md that is not a handle-creation switch (test_gpu_alternate_schedules) is set by one of these."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from limbo_amd import _capi\n"
            "from oracle import binding as OB\n"
            "from tests.test_gpu_parity import _small_parity\n"
            "from tests.util import relerr_norm, MemoLib\n"
            "eng, orc = _capi.load_engine(), MemoLib(OB.load_oracle(), sys.argv[1])  # (the oracle's answers are the same in every child)\n"
            "_small_parity(eng, orc, N=520, seed=7)\n"
            "_small_parity(eng, orc, N=1700, D=4, P=1, seed=8)\n"
            "_small_parity(eng, orc, N=150, D=3, P=1, seed=9)\n"
            "rng = np.random.default_rng(3); X = rng.uniform(0, 1, (1792, 5)); om = np.cos(X.sum(1))[:, None] - 0.1\n"
            "res = []\n"
            "for lib in (eng, orc):\n"
            "    h = lib.handle() if isinstance(lib, MemoLib) else _capi.Handle(lib); h.set_data(X, om)\n"
            "    res.append(h.hp_objective(0, rng.uniform(-0.2, 0.2, 6) * 0 + 0.1, 0.01, optimize_noise=True, want_grad=True)); h.close()\n"
            "assert abs(res[0][0] - res[1][0]) <= 1e-10 * abs(res[1][0]) and relerr_norm(res[0][1], res[1][1]) < 1e-6\n"
            "hs = [_capi.Handle(eng) for _ in range(3)]\n"
            "for q, h in enumerate(hs): h.set_data(X[:700] + 0.01 * q, om[:700]); h.set_kernel(0, np.full(6, 0.05 * q), 0.01)\n"
            "st, bl = _capi.batch_compute(hs), _capi.batch_log_lik(hs)\n"
            "for q, h in enumerate(hs):\n"
            "    o = orc.handle(); o.set_data(X[:700] + 0.01 * q, om[:700]); o.set_kernel(0, np.full(6, 0.05 * q), 0.01); assert o.compute() == 0\n"
            "    assert st[q] == 0 and abs(bl[q] - o.log_lik()) <= 1e-10 * abs(o.log_lik()); o.close(); h.close()\n"
            "hb = [_capi.Handle(eng) for _ in range(2)]\n"
            "for h in hb: h.set_data(X[:1024], om[:1024])\n"
            "thb = np.array([np.full(6, 0.05), np.full(6, -0.05)])\n"
            "lik, grad, st = _capi.batch_hp_objective(hb, 0, thb, np.full(2, 0.01), optimize_noise=False, want_grad=True)\n"
            "for q in range(2):\n"
            "    o = orc.handle(); o.set_data(X[:1024], om[:1024]); r = o.hp_objective(0, thb[q], 0.01, optimize_noise=False, want_grad=True)\n"
            "    assert st[q] == 0 and abs(lik[q] - r[0]) <= 1e-10 * abs(r[0]) and relerr_norm(grad[q], r[1]) < 1e-6; o.close()\n"
            "orc.save()\n"
            "print('child ok')\n") % str(ROOT)
    memo = _switch_memo_dir(tmp_path_factory) / "oracle_memo.pkl"
    r = subprocess.run([sys.executable, "-c", code, str(memo)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr


def test_gpu_above_one_launch_sweep_limit(engine_lib):
    """N = 16448 = 257 blocks of 64: one more than the data-flow sweeps take (every workgroup must be resident),
    so the factorisation's tail, update_alpha and a point query run the per-block sweeps.  Size-independent
    checks: K alpha = obs_mean on sampled rows, mu ~ y and sigma^2 <= 2 (noise + 1e-8) at training points."""
    rng = np.random.default_rng(11)
    N, D = 16448, 3
    X = rng.uniform(0, 1, size=(N, D))
    Y = (np.cos(4 * X).sum(axis=1) + 0.05 * rng.normal(size=N))[:, None]
    om, mean = O.obs_mean_data(Y)
    noise = 0.01
    h = new_gp(engine_lib, O.MATERN52, X, om, np.array([-1.0, 0.0]), noise)
    assert h.compute() == 0
    rows = rng.integers(0, N, size=32)

    def check_alpha(a, rhs):
        d = np.sqrt(((X[rows, None, :] - X[None, :, :]) ** 2).sum(-1)) / np.exp(-1.0)
        t1 = np.sqrt(5.0) * d
        Kr = (1 + t1 + 5.0 * d * d / 3.0) * np.exp(-t1)
        Kr[np.arange(len(rows)), rows] += noise + 1e-8
        assert np.max(np.abs(Kr @ a[:, 0] - rhs[rows, 0])) < 1e-7 * np.max(np.abs(rhs))

    check_alpha(h.get_alpha(), om)
    ll = h.log_lik()
    assert np.isfinite(ll)
    kta, var = h.query_batch(X[rows[:1]])  # a point query
    mu, s2 = O.finish_query(kta, var, mean, noise)
    assert abs(mu[0, 0] - Y[rows[0], 0]) < 1.0 and s2[0] <= 2.0 * (noise + 1e-8)
    om2 = om[::-1].copy()
    h.update_alpha(om2)
    check_alpha(h.get_alpha(), om2)
    h.update_alpha(om)
    assert abs(h.log_lik() - ll) <= 1e-10 * abs(ll)
    h.close()
