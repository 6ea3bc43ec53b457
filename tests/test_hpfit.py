"""limbo_amd/hpfit.py (the Python-side mirror of opt/batched_rprop.hpp): lock-step Rprop restarts walk exactly the iterates of
limbo's sequential Rprop (src/limbo/opt/rprop.hpp:84-144) — checked on the CPU with the oracle's objective (the C restatement
of kernel_lf_opt.hpp:77-92) and against the oracle's own Rprop loop (oracle/gp_oracle.c: orc_kernel_lf_opt_rprop)."""
import numpy as np
import pytest

from limbo_amd import hpfit, synth
from oracle import binding as OB
from oracle import np_oracle as O
from tests.test_gpu_configs import _rprop
from tests.util import new_gp


@pytest.mark.parametrize("on,eps_stop", [(False, 0.0), (True, 0.0), (False, 0.5)])
def test_lockstep_rprop_equals_sequential_rprop(oracle_lib, on, eps_stop):
    rng = np.random.default_rng(5)
    N, D, R = 60, 3, 4
    X = rng.uniform(-2, 2, size=(N, D))
    Y = (np.sin(X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2])[:, None] + 0.05 * rng.normal(size=(N, 1))
    om, _ = synth.obs_mean_data(Y)
    g = new_gp(oracle_lib, O.SE_ARD, X, om, np.zeros(D + 1), 0.01)

    def one(p):
        th, noise = (p[:-1], float(np.exp(2 * p[-1]))) if on else (p, 0.01)
        lik, grad, info = g.hp_objective(O.SE_ARD, th, noise, optimize_noise=on, want_grad=True)
        assert info == 0
        return lik, grad

    def batch(P):
        res = [one(p) for p in P]
        return np.array([r[0] for r in res]), np.array([r[1] for r in res])

    T = D + 1 + (1 if on else 0)
    base = np.concatenate([np.zeros(D + 1), [np.log(np.sqrt(0.01))]]) if on else np.zeros(D + 1)
    inits = base + rng.uniform(-1e-2, 1e-2, size=(R, T))  # parallel_repeater.hpp:88
    inits[0] = base
    trace = []
    bp, bl = hpfit.rprop_lockstep(batch, inits, 25, eps_stop, trace)
    for r in range(R):
        p_seq, l_seq = _rprop(one, inits[r], 25, eps_stop)
        assert np.array_equal(bp[r], p_seq) and bl[r] == l_seq, r
    if eps_stop == 0.0:
        assert len(trace) == 25
        # restart 0 starts where the oracle's own KernelLFOpt + Rprop loop starts: same best-seen parameters
        o = new_gp(oracle_lib, O.SE_ARD, X, om, np.zeros(D + 1), 0.01)
        assert o.compute() == 0
        th_o, ll_o, nev = OB.kernel_lf_opt_rprop(o, optimize_noise=on, iterations=25, eps_stop=0.0)
        assert nev == 25 and np.max(np.abs(bp[0] - th_o)) <= 1e-12 and abs(bl[0] - ll_o) <= 1e-12 * abs(ll_o)
        o.close()
    g.close()
