"""CPU tests: pin the C oracle (oracle/gp_oracle.c) against the reference's own known
answers, the independent numpy/LAPACK restatement and the mpmath goldens.  No GPU."""
import numpy as np
import pytest

from oracle import binding as OB
from oracle import np_oracle as O
from tests import parity_checks as PC
from tests.util import golden_files, new_gp, relerr

import ctypes as C

_dp = C.POINTER(C.c_double)


def _keval(lib, kind, x1, x2, th):
    x1 = np.ascontiguousarray(x1, float)
    x2 = np.ascontiguousarray(x2, float)
    th = np.ascontiguousarray(th, float)
    return lib.cdll.orc_kernel_eval(kind, x1.ctypes.data_as(_dp), x2.ctypes.data_as(_dp), x1.size, th.ctypes.data_as(_dp))


def _kgrad(lib, kind, x1, x2, th):
    x1 = np.ascontiguousarray(x1, float)
    x2 = np.ascontiguousarray(x2, float)
    th = np.ascontiguousarray(th, float)
    g = np.zeros(th.size)
    lib.cdll.orc_kernel_grad(kind, x1.ctypes.data_as(_dp), x2.ctypes.data_as(_dp), x1.size, th.ctypes.data_as(_dp),
                             g.ctypes.data_as(_dp))
    return g


def test_se_ard_known_answers(oracle_lib):
    """src/tests/test_kernel.cpp:196-224 (test_kernel_SE_ARD), k = 0 part."""
    hp = np.zeros(3)
    v1 = np.array([1.0, 1.0])
    assert abs(_keval(oracle_lib, O.SE_ARD, v1, v1, hp) - 1) < 1e-6
    v2 = np.array([0.0, 1.0])
    s1 = _keval(oracle_lib, O.SE_ARD, v1, v2, hp)
    assert abs(s1 - np.exp(-0.5 * float(v1 @ v2))) < 1e-5
    hp[0] = 1
    s2 = _keval(oracle_lib, O.SE_ARD, v1, v2, hp)
    assert s1 < s2


@pytest.mark.parametrize("kind", [O.SE_ARD, O.MATERN52, O.MATERN32, O.EXP])
def test_kernel_grad_fd(oracle_lib, kind):
    """src/tests/test_kernel.cpp:112-194: central FD (e=1e-6) of k wrt every log-hp vs grad(),
    random hp in [-3,3], x in [-5,5]^D, D = 1..10, error < 1e-5."""
    rng = np.random.default_rng(kind)
    e = 1e-6
    for D in range(1, 11):
        nt = D + 1 if kind == O.SE_ARD else 2
        for _ in range(20):
            th = rng.uniform(-3, 3, size=nt)
            x1 = rng.uniform(-5, 5, size=D)
            x2 = rng.uniform(-5, 5, size=D)
            g = _kgrad(oracle_lib, kind, x1, x2, th)
            fd = np.zeros(nt)
            for j in range(nt):
                tp, tm = th.copy(), th.copy()
                tp[j] += e
                tm[j] -= e
                fd[j] = (_keval(oracle_lib, kind, x1, x2, tp) - _keval(oracle_lib, kind, x1, x2, tm)) / (2 * e)
            assert np.linalg.norm(g - fd) < 1e-5


def _keval_n(lib, kind, x1, x2, th):
    x1 = np.ascontiguousarray(x1, float)
    x2 = np.ascontiguousarray(x2, float)
    th = np.ascontiguousarray(th, float)
    return lib.cdll.orc_kernel_eval_n(kind, x1.ctypes.data_as(_dp), x2.ctypes.data_as(_dp), x1.size,
                                      th.ctypes.data_as(_dp), th.size)


def _kgrad_n(lib, kind, x1, x2, th):
    x1 = np.ascontiguousarray(x1, float)
    x2 = np.ascontiguousarray(x2, float)
    th = np.ascontiguousarray(th, float)
    g = np.zeros(th.size)
    lib.cdll.orc_kernel_grad_n(kind, x1.ctypes.data_as(_dp), x2.ctypes.data_as(_dp), x1.size, th.ctypes.data_as(_dp),
                               th.size, g.ctypes.data_as(_dp))
    return g


def test_se_ard_lambda_known_answer(oracle_lib):
    """src/tests/test_kernel.cpp:215-223: with k = 1 and all-zero parameters (Lambda = 0) the value is
    bitwise the k = 0 one."""
    v1, v2 = np.array([1.0, 1.0]), np.array([0.0, 1.0])
    s1 = _keval(oracle_lib, O.SE_ARD, v1, v2, np.zeros(3))
    assert s1 == _keval_n(oracle_lib, O.SE_ARD, v1, v2, np.zeros(2 + 2 * 1 + 1))


@pytest.mark.parametrize("k", [1, 2])
def test_se_ard_lambda_grad_fd(oracle_lib, k):
    """src/tests/test_kernel.cpp:184-188: the same central-FD check with set_k(1), D = 1..10 (k = 2 added;
    the documentation asks for k < D but the code — and the test at D = 1 — do not)."""
    rng = np.random.default_rng(40 + k)
    e = 1e-6
    for D in range(1, 11):
        nt = D + D * k + 1
        if nt > 64:
            continue
        for _ in range(20):
            th = rng.uniform(-3, 3, size=nt)
            x1 = rng.uniform(-5, 5, size=D)
            x2 = rng.uniform(-5, 5, size=D)
            g = _kgrad_n(oracle_lib, O.SE_ARD, x1, x2, th)
            fd = np.zeros(nt)
            for j in range(nt):
                tp, tm = th.copy(), th.copy()
                tp[j] += e
                tm[j] -= e
                fd[j] = (_keval_n(oracle_lib, O.SE_ARD, x1, x2, tp) - _keval_n(oracle_lib, O.SE_ARD, x1, x2, tm)) / (2 * e)
            assert np.linalg.norm(g - fd) < 1e-5


def test_se_ard_lambda_gp_vs_numpy(oracle_lib):
    """K, L, alpha, log-lik, gradient (with noise) and queries of a GP with a D x 2 Lambda against the
    numpy/LAPACK restatement (einsum over M = Lambda Lambda^T + diag(ell^-2))."""
    rng = np.random.default_rng(77)
    N, D, k = 61, 4, 2
    X = rng.uniform(-2, 2, size=(N, D))
    Y = rng.normal(size=(N, 2))
    th = np.concatenate([rng.uniform(-0.3, 0.5, size=D), rng.uniform(-0.6, 0.6, size=D * k), [0.2]])
    h = new_gp(oracle_lib, O.SE_ARD, X, Y, th, 0.05)
    assert h.compute() == 0
    K, L, al = O.gp_fit(O.SE_ARD, X, Y, th, 0.05)
    assert relerr(h.get_K(), K, floor=1e-30) < 1e-12
    assert relerr(h.get_L(), L, floor=1e-30) < 1e-10
    assert relerr(h.get_alpha(), al) < 1e-9
    assert abs(h.log_lik() - O.log_lik(L, Y, al)) < 1e-9 * abs(O.log_lik(L, Y, al))
    g = h.log_lik_grad(True)
    gn = O.log_lik_grad(O.SE_ARD, X, th, 0.05, L, al, True)
    assert g.size == th.size + 1 and relerr(g, gn, floor=1e-6) < 1e-8
    Xq = rng.uniform(-2, 2, size=(9, D))
    kta, var = h.query_batch(Xq)
    ktan, varn = O.query(O.SE_ARD, X, th, L, al, Xq)
    assert relerr(kta, ktan, floor=1e-9) < 1e-9 and relerr(var, varn, floor=1e-9) < 1e-8
    h.close()


@pytest.mark.parametrize("kind", [O.SE_ARD, O.MATERN52, O.MATERN32, O.EXP])
def test_kernel_matrix_vs_numpy(oracle_lib, kind):
    rng = np.random.default_rng(100 + kind)
    X = rng.uniform(-2, 2, size=(57, 5))
    nt = 6 if kind == O.SE_ARD else 2
    th = rng.uniform(-1, 1, size=nt)
    h = new_gp(oracle_lib, kind, X, np.zeros((57, 1)), th, 0.03)
    K = h.get_K()
    Kn = O.kernel_matrix(kind, X, th, 0.03)
    assert relerr(K, Kn, floor=1e-30) < 1e-13
    assert np.array_equal(K, K.T)
    h.close()


@pytest.mark.parametrize("path", golden_files("mp_"), ids=lambda p: p.stem)
def test_oracle_vs_mpmath_golden(oracle_lib, path):
    PC.check_against_mp_golden(oracle_lib, path)


@pytest.mark.parametrize("path", golden_files("np_"), ids=lambda p: p.stem)
def test_oracle_vs_lapack_golden(oracle_lib, path):
    PC.check_against_np_golden(oracle_lib, path)


@pytest.mark.parametrize("dup", [False, True])
def test_oracle_incremental_vs_full(oracle_lib, dup):
    PC.check_incremental_vs_full(oracle_lib, dup=dup)


def test_oracle_add_sample_from_empty(oracle_lib):
    PC.check_add_sample_from_empty(oracle_lib)


@pytest.mark.parametrize("kind,on", [(O.SE_ARD, False), (O.SE_ARD, True), (O.MATERN52, True)])
def test_oracle_grad_fd(oracle_lib, kind, on):
    PC.check_grad_fd(oracle_lib, kind, on)


@pytest.mark.parametrize("kind,on", [(O.SE_ARD, False), (O.SE_ARD, True), (O.MATERN52, True)])
def test_oracle_loo_grad_fd(oracle_lib, kind, on):
    PC.check_loo_grad_fd(oracle_lib, kind, on)


def test_oracle_loo_collapsed_form(oracle_lib):
    """The one-product form of the LOO gradient used on the device (grad.hip header) equals the
    reference's per-parameter Zeta products (gp.hpp:380-389), evaluated here in numpy from the
    oracle's K^-1 and alpha."""
    rng = np.random.default_rng(77)
    N, D, P = 57, 3, 2
    X = rng.uniform(-1, 1, size=(N, D))
    Y = np.stack([np.sin(X.sum(axis=1) * (p + 1)) for p in range(P)], axis=1)
    om, _ = O.obs_mean_data(Y)
    th = np.array([0.2, -0.3, 0.1, 0.05])
    h = new_gp(oracle_lib, O.SE_ARD, X, om, th, 0.02)
    assert h.compute() == 0
    ref = h.log_loo_cv_grad(True)
    Ki, al = h.get_Kinv(), h.get_alpha()
    kappa = np.diag(Ki)
    u = Ki @ (al / kappa[:, None])
    c = (0.5 * (1 + al ** 2 / kappa[:, None]) / kappa[:, None]).sum(axis=1)
    W = 0.5 * (u @ al.T + al @ u.T) - (Ki * c[None, :]) @ Ki
    G = O.kernel_grad_tensor(O.SE_ARD, X, th)
    got = np.append((G * W[None]).sum(axis=(1, 2)), 2 * 0.02 * np.trace(W))
    assert np.linalg.norm(got - ref) < 1e-10 * max(1.0, np.linalg.norm(ref))
    h.close()


def test_oracle_update_alpha_and_clone(oracle_lib):
    PC.check_update_alpha_and_clone(oracle_lib)


def test_oracle_host_K(oracle_lib):
    PC.check_host_K(oracle_lib)


def test_oracle_not_pd(oracle_lib):
    PC.check_not_pd(oracle_lib)


def test_oracle_interpolation_and_prior(oracle_lib):
    """test_gp.cpp:448-511, :669-758: mu within 1 of y and sigma^2 <= 2(noise+1e-8) at the
    training points; far away sigma^2 -> sigma_f^2 + noise."""
    rng = np.random.default_rng(21)
    X = rng.uniform(0, 1, size=(30, 2))
    Y = np.sin(6 * X[:, :1])
    om, mean = O.obs_mean_data(Y)
    th = np.array([-1.0, -1.0, 0.0])
    h = new_gp(oracle_lib, O.SE_ARD, X, om, th, 0.01)
    h.compute()
    kta, var = h.query_batch(X)
    mu, s2 = O.finish_query(kta, var, mean, 0.01)
    assert np.max(np.abs(mu - Y)) < 1.0
    assert np.all(s2 <= 2 * (0.01 + 1e-8))
    far = np.full((1, 2), 50.0)
    kta, var = h.query_batch(far)
    mu, s2 = O.finish_query(kta, var, mean, 0.01)
    assert abs(s2[0] - (1.0 + 0.01)) < 0.01
    h.close()


def test_oracle_rprop_improves(oracle_lib):
    """KernelLFOpt with Rprop (kernel_lf_opt.hpp:60-69, rprop.hpp:84-144): the returned
    hyper-parameters are the best seen and are no worse than the start."""
    X, Y = O.make_problem("c1", N=60)
    om, _ = O.obs_mean_data(Y)
    h = new_gp(oracle_lib, O.SE_ARD, X, om, np.zeros(3), 0.01)
    h.compute()
    ll0 = h.log_lik()
    th, ll, nev = OB.kernel_lf_opt_rprop(h, optimize_noise=True, iterations=50, eps_stop=1e-2)
    assert nev <= 50
    assert ll >= ll0
    assert abs(h.log_lik() - ll) < 1e-12 * abs(ll)
    h.close()


def _np_sparsify(X, max_points):
    """numpy restatement of sparsified_gp.hpp:124-183 (serial order), independent of the C oracle."""
    idx = list(range(X.shape[0]))
    Dm = np.sqrt(((X[:, None, :] - X[None, :, :]) ** 2).sum(axis=2))
    D = X.shape[1]
    while len(idx) > max_points:
        sub = Dm[np.ix_(idx, idx)]
        best, bi = np.inf, -1
        for a in range(len(idx)):
            nb = np.sort(np.delete(sub[a], a))[:D]
            s = 0.0
            for v in nb:
                s += v
            if s < best:
                best, bi = s, a
        idx.pop(bi)
    return np.array(idx)


@pytest.mark.parametrize("N,D,mp", [(60, 2, 20), (45, 3, 44), (30, 4, 30), (80, 1, 7)])
def test_oracle_sparsify(oracle_lib, N, D, mp):
    from limbo_amd import _capi
    rng = np.random.default_rng(N + D)
    X = rng.uniform(0, 1, size=(N, D))
    X[: N // 3] = 0.5 + 0.05 * rng.normal(size=(N // 3, D))  # a dense cluster that must be thinned first
    keep = _capi.sparsify(oracle_lib, X, mp)
    assert len(keep) == min(N, mp) and np.all(np.diff(keep) > 0)
    # the numpy restatement sums the squares in a different order (last-ulp differences in the
    # distances); on continuous random data the selection is the same
    assert np.array_equal(keep, _np_sparsify(X, mp))
    if N > mp:  # the cluster lost proportionally more points than the background
        assert (keep < N // 3).mean() < (N // 3) / N
