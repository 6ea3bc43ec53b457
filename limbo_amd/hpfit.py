"""KernelLFOpt + Rprop for R restarts advanced in lock-step — the Python-side mirror of the drop-in's
`opt/batched_rprop.hpp` + `model/gp/kernel_lf_opt.hpp`, for the test and benchmark harness.

limbo runs the restarts of a hyper-parameter fit as independent tasks (src/limbo/opt/parallel_repeater.hpp:84-105), each a
sequential Rprop (src/limbo/opt/rprop.hpp:84-144) whose every objective evaluation is
KernelLFOptimization::operator() (src/limbo/model/gp/kernel_lf_opt.hpp:77-92).  On the device R such evaluations are ONE
launch sequence (gpe_batch_hp_objective) — if the R optimisers ask for them at the same time: iteration i of every restart
is one batched call.  Member r's iterates are exactly those of opt::Rprop started from inits[r].
"""
from __future__ import annotations

import numpy as np

from . import _capi

# rprop.hpp:89-93
DELTA0, DELTA_MIN, DELTA_MAX, ETA_MINUS, ETA_PLUS = 0.1, 1e-6, 50.0, 0.5, 1.2


def rprop_lockstep(objective_batch, inits, iterations, eps_stop=0.0, trace=None):
    """objective_batch(params (R x T)) -> (lik (R,), grad (R x T)); maximises.  Returns (best_params (R x T), best (R,)):
    the best point SEEN per restart (rprop.hpp:115-118,143).  A restart whose gradient norm falls below eps_stop stops
    (:139-141) and is evaluated where it stands from then on (ignored).  trace: list that receives (params, lik, grad) per
    iteration."""
    params = np.array(inits, dtype=np.float64, copy=True)
    R, T = params.shape
    step = np.full((R, T), DELTA0)
    prev = np.zeros((R, T))
    best = np.full(R, -np.inf)
    best_params = params.copy()
    live = np.ones(R, dtype=bool)
    for _ in range(iterations):
        if not live.any():
            break
        lik, grad = objective_batch(params)
        if trace is not None:
            trace.append((params.copy(), np.array(lik, copy=True), np.array(grad, copy=True)))
        for r in range(R):
            if not live[r]:
                continue
            if lik[r] > best[r]:
                best[r] = lik[r]
                best_params[r] = params[r]
            g = -np.asarray(grad[r], dtype=np.float64)
            s = prev[r] * g
            up, dn = s > 0, s < 0
            step[r, up] = np.minimum(step[r, up] * ETA_PLUS, DELTA_MAX)
            step[r, dn] = np.maximum(step[r, dn] * ETA_MINUS, DELTA_MIN)
            g[dn] = 0.0
            params[r] -= np.sign(g) * step[r]
            prev[r] = g
            if np.linalg.norm(g) < eps_stop:
                live[r] = False
    return best_params, best


def kernel_lf_opt_lockstep(handles, kind, inits, noise=0.01, optimize_noise=False, iterations=300, eps_stop=0.0, trace=None):
    """A KernelLFOpt fit with len(handles) restarts on device clones of one GP (same data on every handle), every
    iteration ONE gpe_batch_hp_objective.  inits: (R x n_params), log-space (with optimize_noise the last entry is
    log(sqrt(noise)): kernel.hpp:99-113).  Returns (best_params, best_lik)."""
    R = len(handles)

    def objective(p):
        if optimize_noise:
            th, nz = p[:, :-1], np.exp(2.0 * p[:, -1])
        else:
            th, nz = p, np.full(R, noise)
        lik, grad, st = _capi.batch_hp_objective(handles, kind, th, nz, optimize_noise=optimize_noise, want_grad=True)
        if any(s != 0 for s in st):
            raise _capi.EngineError(f"batched objective: status {st}")
        return lik, grad

    return rprop_lockstep(objective, inits, iterations, eps_stop, trace)
