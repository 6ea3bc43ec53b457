"""ctypes loaders of the CHECKERS — TEST INFRASTRUCTURE ONLY.

Imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from limbo_amd/.

* ``liboracle.so`` (prefix ``orc_``): the C restatement of the reference's GP path (gp_oracle.c).  It exports
  the signatures of include/gpe.h, so the product's ``Handle`` class drives it unchanged.
* ``_ref/libref.so`` (prefix ``ref_``): the UNMODIFIED reference (limbo::model::GP from /root/reference/src)
  compiled against the Eigen/Boost stand-ins of oracle/ref_build/ — the arithmetic underneath is the
  stand-in's, every semantic decision is the reference's own source.  Built by ``make -C oracle/ref_build``
  where /root/reference exists (this container); the GPU box only ever sees the prebuilt file.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from limbo_amd import _capi

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "liboracle.so"
REF_SO = HERE / "_ref" / "libref.so"
REFERENCE_SRC = Path("/root/reference/src")

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_libs = {}


def load_oracle() -> _capi.Lib:
    if "orc" not in _libs:
        if not ORACLE_SO.exists():
            subprocess.check_call(["make", "-C", str(HERE)])
        lib = _capi.Lib(ORACLE_SO, "orc_")
        cd = lib.cdll
        cd.orc_kernel_lf_opt_rprop.argtypes = [_vp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.POINTER(C.c_int)]
        cd.orc_kernel_lf_opt_rprop.restype = C.c_int
        cd.orc_kernel_eval.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp]
        cd.orc_kernel_eval.restype = C.c_double
        cd.orc_kernel_grad.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        cd.orc_kernel_grad.restype = None
        cd.orc_kernel_eval_n.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, C.c_int]
        cd.orc_kernel_eval_n.restype = C.c_double
        cd.orc_kernel_grad_n.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, C.c_int, _dp]
        cd.orc_kernel_grad_n.restype = None
        _libs["orc"] = lib
    return _libs["orc"]


def kernel_lf_opt_rprop(h: _capi.Handle, optimize_noise=False, iterations=300, eps_stop=0.0):
    """KernelLFOpt + Rprop on an oracle handle (model/gp/kernel_lf_opt.hpp:60-92, opt/rprop.hpp:84-144)."""
    assert h.lib.prefix == "orc_"
    n = h.n_theta + (1 if optimize_noise else 0)
    th = np.zeros(n)
    lik = C.c_double()
    ne = C.c_int()
    rc = h.lib.cdll.orc_kernel_lf_opt_rprop(h._h, int(optimize_noise), iterations, eps_stop,
                                            th.ctypes.data_as(_dp), C.byref(lik), C.byref(ne))
    h._chk(rc, "kernel_lf_opt_rprop")
    return th, lik.value, ne.value


# ------------------------------------------------------------------------------------------------
# the reference itself
# ------------------------------------------------------------------------------------------------
def ref_available() -> bool:
    return REF_SO.exists() or REFERENCE_SRC.exists()


def load_ref():
    if "ref" not in _libs:
        if REFERENCE_SRC.exists():  # (re)build where the reference sources are; a no-op when up to date
            subprocess.check_call(["make", "-s", "-C", str(HERE / "ref_build")])
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} is missing and /root/reference is not here to build it")
        cd = C.CDLL(str(REF_SO), mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_LOCAL", 0))
        i64 = C.c_int64
        sig = {
            "ref_version": ([], C.c_char_p),
            "ref_set_statics": ([C.c_double, C.c_int, C.c_int, C.c_double], None),
            "ref_set_rprop": ([C.c_int, C.c_double], None),
            "ref_create": ([C.c_int, C.c_int, C.c_int, C.c_int], _vp),
            "ref_destroy": ([_vp], None),
            "ref_compute": ([_vp, _dp, _dp, i64, C.c_int, C.c_int], None),
            "ref_add_sample": ([_vp, _dp, C.c_int, _dp, C.c_int], None),
            "ref_recompute": ([_vp, C.c_int, C.c_int], None),
            "ref_set_kernel_h_params": ([_vp, _dp, C.c_int], None),
            "ref_kernel_h_params_size": ([_vp], C.c_int),
            "ref_get_kernel_h_params": ([_vp, _dp], None),
            "ref_set_mean_h_params": ([_vp, _dp, C.c_int], None),
            "ref_mean_h_params_size": ([_vp], C.c_int),
            "ref_get_mean_h_params": ([_vp, _dp], None),
            "ref_noise": ([_vp], C.c_double),
            "ref_query": ([_vp, _dp, i64, C.c_int, _dp, _dp], None),
            "ref_mu": ([_vp, _dp, C.c_int, _dp], None),
            "ref_sigma": ([_vp, _dp, C.c_int], C.c_double),
            "ref_log_lik": ([_vp], C.c_double),
            "ref_kernel_grad_log_lik": ([_vp, _dp], None),
            "ref_mean_grad_log_lik": ([_vp, _dp], None),
            "ref_log_loo_cv": ([_vp], C.c_double),
            "ref_kernel_grad_log_loo_cv": ([_vp, _dp], None),
            "ref_compute_inv_kernel": ([_vp], None),
            "ref_inv_kernel_computed": ([_vp], C.c_int),
            "ref_nb_samples": ([_vp], i64),
            "ref_get_matrix": ([_vp, C.c_int, _dp], None),
            "ref_optimize_hyperparams": ([_vp, C.c_int], None),
            "ref_kernel_eval": ([_vp, _dp, _dp, C.c_int, C.c_int, C.c_int], C.c_double),
            "ref_kernel_grad": ([_vp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp], None),
            "ref_save": ([_vp, C.c_char_p, C.c_int], None),
            "ref_acqui": ([_vp, C.c_int, C.c_int, _dp, i64, C.c_int, _dp], None),
            "ref_load": ([_vp, C.c_char_p, C.c_int, C.c_int], None),
        }
        for name, (args, res) in sig.items():
            f = getattr(cd, name)
            f.argtypes = args
            f.restype = res
        _libs["ref"] = cd
    return _libs["ref"]


def _c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


MEAN_DATA, MEAN_NULL, MEAN_CONSTANT = 0, 1, 2
OPT_KERNEL_LF, OPT_KERNEL_LOO, OPT_MEAN_LF, OPT_KERNEL_MEAN_LF = 0, 1, 2, 3


class RefGP:
    """limbo::model::GP<Params, Kernel, Mean, NoLFOpt> of the reference, by value."""

    def __init__(self, kind, D, P=1, mean=MEAN_DATA, noise=0.01, optimize_noise=False, k_lambda=0, constant=1.0):
        self.lib = load_ref()
        self.lib.ref_set_statics(float(noise), int(optimize_noise), int(k_lambda), float(constant))
        self.h = self.lib.ref_create(int(kind), int(mean), int(D), int(P))
        assert self.h, "unknown kernel/mean kind"
        self.D, self.P = D, P

    def close(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def N(self):
        return int(self.lib.ref_nb_samples(self.h))

    def set_h_params(self, p):
        p = _c(p)
        assert p.size == self.lib.ref_kernel_h_params_size(self.h), (p.size, self.lib.ref_kernel_h_params_size(self.h))
        self.lib.ref_set_kernel_h_params(self.h, _p(p), p.size)

    def h_params(self):
        p = np.zeros(self.lib.ref_kernel_h_params_size(self.h))
        self.lib.ref_get_kernel_h_params(self.h, _p(p))
        return p

    def set_mean_h_params(self, p):
        p = _c(p)
        self.lib.ref_set_mean_h_params(self.h, _p(p), p.size)

    def mean_h_params(self):
        p = np.zeros(self.lib.ref_mean_h_params_size(self.h))
        self.lib.ref_get_mean_h_params(self.h, _p(p))
        return p

    def noise(self):
        return float(self.lib.ref_noise(self.h))

    def compute(self, X, Y):
        X = _c(X)
        Y = _c(np.asarray(Y, float).reshape(X.shape[0], -1))
        self.lib.ref_compute(self.h, _p(X), _p(Y), X.shape[0], X.shape[1], Y.shape[1])

    def add_sample(self, x, y):
        x, y = _c(x), _c(np.atleast_1d(y))
        self.lib.ref_add_sample(self.h, _p(x), x.size, _p(y), y.size)

    def recompute(self, update_obs_mean=True, update_full_kernel=True):
        self.lib.ref_recompute(self.h, int(update_obs_mean), int(update_full_kernel))

    def acqui(self, which, Xq, iteration=0):
        """limbo's own acquisition functors over this model (acqui/ucb.hpp, gp_ucb.hpp, ei.hpp): 0 UCB, 1 GP_UCB, 2 EI; first output."""
        Xq = _c(np.atleast_2d(Xq))
        out = np.zeros(Xq.shape[0])
        self.lib.ref_acqui(self.h, int(which), int(iteration), _p(Xq), Xq.shape[0], Xq.shape[1], _p(out))
        return out

    def save(self, directory, binary):
        """GP::save<TextArchive | BinaryArchive>(directory) of the reference (gp.hpp:439-460): real files."""
        self.lib.ref_save(self.h, str(directory).encode(), int(binary))

    def load(self, directory, binary, recompute=True):
        """GP::load<TextArchive | BinaryArchive>(directory, recompute) of the reference (gp.hpp:462-511)."""
        self.lib.ref_load(self.h, str(directory).encode(), int(binary), int(recompute))

    def query(self, Xq):
        Xq = _c(np.atleast_2d(Xq))
        mu = np.zeros((Xq.shape[0], self.P))
        s2 = np.zeros(Xq.shape[0])
        self.lib.ref_query(self.h, _p(Xq), Xq.shape[0], Xq.shape[1], _p(mu), _p(s2))
        return mu, s2

    def mu(self, x):
        x = _c(x)
        m = np.zeros(self.P)
        self.lib.ref_mu(self.h, _p(x), x.size, _p(m))
        return m

    def sigma(self, x):
        x = _c(x)
        return float(self.lib.ref_sigma(self.h, _p(x), x.size))

    def log_lik(self):
        return float(self.lib.ref_log_lik(self.h))

    def kernel_grad_log_lik(self):
        g = np.zeros(self.lib.ref_kernel_h_params_size(self.h))
        self.lib.ref_kernel_grad_log_lik(self.h, _p(g))
        return g

    def mean_grad_log_lik(self):
        g = np.zeros(self.lib.ref_mean_h_params_size(self.h))
        self.lib.ref_mean_grad_log_lik(self.h, _p(g))
        return g

    def log_loo_cv(self):
        return float(self.lib.ref_log_loo_cv(self.h))

    def kernel_grad_log_loo_cv(self):
        g = np.zeros(self.lib.ref_kernel_h_params_size(self.h))
        self.lib.ref_kernel_grad_log_loo_cv(self.h, _p(g))
        return g

    def compute_inv_kernel(self):
        self.lib.ref_compute_inv_kernel(self.h)

    def inv_kernel_computed(self):
        return bool(self.lib.ref_inv_kernel_computed(self.h))

    def _mat(self, which, cols):
        n = self.N
        A = np.zeros((n, cols), order="F")
        self.lib.ref_get_matrix(self.h, which, _p(A))
        return A

    def matrixL(self):
        return self._mat(0, self.N)

    def alpha(self):
        return self._mat(1, self.P)

    def obs_mean(self):
        return self._mat(2, self.P)

    def inv_kernel(self):
        return self._mat(3, self.N)

    def kernel_matrix(self):
        return self._mat(4, self.N)

    def mean_vector(self):
        return self._mat(5, self.P)

    def optimize_hyperparams(self, which=OPT_KERNEL_LF, iterations=300, eps_stop=0.0):
        self.lib.ref_set_rprop(int(iterations), float(eps_stop))
        self.lib.ref_optimize_hyperparams(self.h, int(which))

    def kernel_eval(self, a, b, i=-1, j=-2):
        a, b = _c(a), _c(b)
        return float(self.lib.ref_kernel_eval(self.h, _p(a), _p(b), a.size, int(i), int(j)))

    def kernel_grad(self, a, b, i=-1, j=-2):
        a, b = _c(a), _c(b)
        g = np.zeros(self.lib.ref_kernel_h_params_size(self.h))
        self.lib.ref_kernel_grad(self.h, _p(a), _p(b), a.size, int(i), int(j), _p(g))
        return g
