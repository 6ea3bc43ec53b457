"""ctypes binding of the C-ABI declared in include/gpe.h.

``libgpengine.so`` (prefix ``gpe_``) is the product: hand-written HIP for gfx950.
:func:`load_engine` raises if it is missing — there is no CPU fallback, and this module
knows of no other implementation.  (``Lib`` takes the symbol prefix as an argument so that
the test infrastructure under ``oracle/`` can bind its checker, which exports the same
signatures, with the same ``Handle`` class; that loader lives in ``oracle/binding.py``.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
ROOT = _PKG.parent
ENGINE_SO = _PKG / "libgpengine.so"

KERNEL_SE_ARD, KERNEL_MATERN52, KERNEL_MATERN32, KERNEL_EXP, KERNEL_HOST_K = range(5)
KERNEL_NAMES = {"se_ard": 0, "matern52": 1, "matern32": 2, "exp": 3, "host_k": 4}

PH_KERNEL_BUILD, PH_POTRF_PANEL, PH_POTRF_UPDATE, PH_SOLVE, PH_LOGLIK, PH_INV, PH_GRAD, PH_QUERY, PH_POTRF_TALL, PH_POTRF_TAIL = range(10)
PH_COUNT = 10
PHASE_NAMES = ["kernel_build", "potrf_panel", "potrf_update", "solve", "loglik", "inv", "grad", "query", "potrf_tall", "potrf_tail"]

_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_vp = C.c_void_p


class EngineError(RuntimeError):
    pass


def _sig(lib, prefix):
    """Declare argtypes/restype for every symbol of include/gpe.h (required symbols
    must exist; a missing one raises AttributeError -> surfaced by tests)."""
    S = {
        "create": [C.c_int, C.POINTER(_vp)],
        "clone": [_vp, C.POINTER(_vp)],
        "clone_to": [_vp, C.c_int, C.POINTER(_vp)],
        "device_count": [C.POINTER(C.c_int)],
        "get_device": [_vp, C.POINTER(C.c_int)],
        "destroy": [_vp],
        "set_data": [_vp, _dp, _i64, C.c_int, _dp, C.c_int],
        "set_data_device": [_vp, _vp, _i64, C.c_int, _vp, C.c_int],
        "set_kernel": [_vp, C.c_int, _dp, C.c_int, C.c_double],
        "set_K_host": [_vp, _dp, _i64],
        "compute": [_vp],
        "update_alpha": [_vp, _dp],
        "add_sample": [_vp, _dp, C.c_int, _dp, C.c_int],
        "log_lik": [_vp, _dp],
        "compute_inv_kernel": [_vp],
        "log_lik_grad": [_vp, _dp, C.c_int, C.c_int],
        "sparsify": [C.c_int, _dp, _i64, C.c_int, _i64, C.POINTER(_i64), C.POINTER(_i64)],
        "log_loo_cv": [_vp, _dp],
        "log_loo_cv_grad": [_vp, _dp, C.c_int, C.c_int],
        "get_loo_weights": [_vp, _dp, _i64],
        "hp_objective": [_vp, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp],
        "query_batch": [_vp, _dp, _i64, _dp, _dp],
        "query_batch_cross": [_vp, _dp, _i64, _dp, _dp],
        "set_obs_mean": [_vp, _dp],
        "nb_samples": [_vp, C.POINTER(_i64)],
        "get_L": [_vp, _dp, _i64],
        "set_L": [_vp, _dp, _i64],
        "get_alpha": [_vp, _dp],
        "set_alpha": [_vp, _dp],
        "get_Kinv": [_vp, _dp, _i64],
        "get_K": [_vp, _dp, _i64],
        "batch_compute": [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int)],
        "batch_log_lik": [C.POINTER(_vp), C.c_int, _dp],
        "batch_hp_objective": [C.POINTER(_vp), C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.POINTER(C.c_int)],
        "synchronize": [_vp],
    }
    for name, args in S.items():
        f = getattr(lib, prefix + name)
        f.argtypes = args
        f.restype = C.c_int
    for name in ("last_error",):
        f = getattr(lib, prefix + name)
        f.argtypes = [_vp]
        f.restype = C.c_char_p
    f = getattr(lib, prefix + "version")
    f.argtypes = []
    f.restype = C.c_char_p
    if prefix == "gpe_":
        G = {
            "get_stream": [_vp, C.POINTER(_vp)],
            "set_profiling": [_vp, C.c_int],
            "get_phase_ms": [_vp, _dp, C.POINTER(_i64), _dp, C.c_int],
            "reset_phase_ms": [_vp],
            "flow_retries": [_vp, C.POINTER(_i64)],
            "handover_reruns": [_vp, C.POINTER(_i64)],
            "small_calls": [_vp, C.POINTER(_i64)],
            "mfma_f64_peak": [C.c_int, _dp],
            "trace": [C.c_int],
            "trace_dump": [C.c_char_p],
            "hbm_stream_peak": [C.c_int, _dp],
            "epoch": [_vp, C.POINTER(C.c_uint64)],
            "xproc_waits": [C.POINTER(_i64)],
        }
        for name, args in G.items():
            f = getattr(lib, prefix + name)
            f.argtypes = args
            f.restype = C.c_int


class Lib:
    def __init__(self, path, prefix):
        self.path = str(path)
        self.prefix = prefix
        self.cdll = C.CDLL(self.path, mode=getattr(os, "RTLD_NOW", 2) | getattr(os, "RTLD_LOCAL", 0))
        _sig(self.cdll, prefix)

    def fn(self, name):
        return getattr(self.cdll, self.prefix + name)


_libs = {}


def load_engine() -> Lib:
    """The HIP library.  Fails loudly when it has not been built (no CPU fallback)."""
    if "gpe" not in _libs:
        if not ENGINE_SO.exists():
            raise EngineError(
                f"{ENGINE_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _libs["gpe"] = Lib(ENGINE_SO, "gpe_")
    return _libs["gpe"]


def _d(a):
    return a.ctypes.data_as(_dp)


def _c(a, order="C"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["C" if order == "C" else "F", "A"])


class Handle:
    """One GP behind the C-ABI.  Method names follow include/gpe.h."""

    def __init__(self, lib: Lib, device: int = 0, _h=None):
        self.lib = lib
        self.N = 0
        self.D = 0
        self.P = 0
        self.n_theta = 0
        if _h is None:
            h = _vp()
            self._chk(lib.fn("create")(device, C.byref(h)), None)
            self._h = h
        else:
            self._h = _h

    # -- plumbing
    def _chk(self, rc, what="call"):
        if rc < 0:
            msg = ""
            if what is not None and getattr(self, "_h", None):
                m = self.lib.fn("last_error")(self._h)
                msg = m.decode() if m else ""
            raise EngineError(f"{self.lib.prefix}{what} failed: status {rc} {msg}")
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fn("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self, device=None) -> "Handle":
        h = _vp()
        if device is None:
            self._chk(self.lib.fn("clone")(self._h, C.byref(h)), "clone")
        else:
            self._chk(self.lib.fn("clone_to")(self._h, int(device), C.byref(h)), "clone_to")
        o = Handle(self.lib, _h=h)
        o.N, o.D, o.P, o.n_theta = self.N, self.D, self.P, self.n_theta
        return o

    def device(self) -> int:
        d = C.c_int()
        self._chk(self.lib.fn("get_device")(self._h, C.byref(d)), "get_device")
        return d.value

    def epoch(self) -> int:
        """Moves with every call that can change what a query answers (include/gpe.h: gpe_epoch)."""
        e = C.c_uint64()
        self._chk(self.lib.fn("epoch")(self._h, C.byref(e)), "epoch")
        return e.value

    def flow_retries(self) -> int:
        n = _i64()
        self._chk(self.lib.fn("flow_retries")(self._h, C.byref(n)), "flow_retries")
        return n.value

    def handover_reruns(self) -> int:
        n = _i64()
        self._chk(self.lib.fn("handover_reruns")(self._h, C.byref(n)), "handover_reruns")
        return n.value

    def small_calls(self) -> int:
        n = _i64()
        self._chk(self.lib.fn("small_calls")(self._h, C.byref(n)), "small_calls")
        return n.value

    # -- data
    def set_data(self, X, obs_mean):
        X = _c(X)
        om = np.asarray(obs_mean, dtype=np.float64)
        if om.ndim == 1:
            om = om[:, None]
        om = _c(om, "F")
        self.N, self.D = X.shape
        self.P = om.shape[1]
        assert om.shape[0] == self.N
        self._chk(self.lib.fn("set_data")(self._h, _d(X), self.N, self.D, _d(om), self.P), "set_data")

    def set_data_device(self, dX_ptr: int, N: int, D: int, dom_ptr: int, P: int):
        self.N, self.D, self.P = N, D, P
        self._chk(self.lib.fn("set_data_device")(self._h, _vp(dX_ptr), N, D, _vp(dom_ptr), P), "set_data_device")

    def set_kernel(self, kind, log_theta, noise):
        kind = KERNEL_NAMES.get(kind, kind)
        th = _c(log_theta)
        self.n_theta = th.size
        self.kind = kind
        self._chk(self.lib.fn("set_kernel")(self._h, kind, _d(th), th.size, float(noise)), "set_kernel")

    def set_K_host(self, K):
        K = _c(K, "F")
        self._chk(self.lib.fn("set_K_host")(self._h, _d(K), K.shape[0]), "set_K_host")

    # -- hot path
    def compute(self) -> int:
        return self._chk(self.lib.fn("compute")(self._h), "compute")

    def update_alpha(self, obs_mean=None):
        if obs_mean is None:
            return self._chk(self.lib.fn("update_alpha")(self._h, None), "update_alpha")
        om = np.asarray(obs_mean, dtype=np.float64)
        if om.ndim == 1:
            om = om[:, None]
        om = _c(om, "F")
        return self._chk(self.lib.fn("update_alpha")(self._h, _d(om)), "update_alpha")

    def add_sample(self, x, obs_mean):
        x = _c(x)
        om = np.asarray(obs_mean, dtype=np.float64)
        if om.ndim == 1:
            om = om[:, None]
        om = _c(om, "F")
        if self.N == 0:
            self.D = x.size
            self.P = om.shape[1]
        assert om.shape[0] == self.N + 1
        rc = self._chk(self.lib.fn("add_sample")(self._h, _d(x), x.size, _d(om), om.shape[1]), "add_sample")
        self.N += 1
        return rc

    def log_lik(self) -> float:
        out = C.c_double()
        self._chk(self.lib.fn("log_lik")(self._h, C.byref(out)), "log_lik")
        return out.value

    def compute_inv_kernel(self):
        return self._chk(self.lib.fn("compute_inv_kernel")(self._h), "compute_inv_kernel")

    def log_lik_grad(self, optimize_noise=False):
        n = self.n_theta + (1 if optimize_noise else 0)
        g = np.zeros(n)
        self._chk(self.lib.fn("log_lik_grad")(self._h, _d(g), n, int(optimize_noise)), "log_lik_grad")
        return g

    def log_loo_cv(self) -> float:
        out = C.c_double()
        self._chk(self.lib.fn("log_loo_cv")(self._h, C.byref(out)), "log_loo_cv")
        return out.value

    def log_loo_cv_grad(self, optimize_noise=False):
        n = self.n_theta + (1 if optimize_noise else 0)
        g = np.zeros(n)
        self._chk(self.lib.fn("log_loo_cv_grad")(self._h, _d(g), n, int(optimize_noise)), "log_loo_cv_grad")
        return g

    def get_loo_weights(self):
        W = np.zeros((self.N, self.N), order="F")
        self._chk(self.lib.fn("get_loo_weights")(self._h, _d(W), self.N), "get_loo_weights")
        return W

    def hp_objective(self, kind, log_theta, noise, optimize_noise=False, want_grad=True):
        kind = KERNEL_NAMES.get(kind, kind)
        th = _c(log_theta)
        self.n_theta = th.size
        lik = C.c_double()
        g = np.zeros(th.size + (1 if optimize_noise else 0))
        rc = self._chk(self.lib.fn("hp_objective")(self._h, kind, _d(th), th.size, float(noise),
                                                   int(optimize_noise), int(want_grad), C.byref(lik),
                                                   _d(g) if want_grad else None), "hp_objective")
        return (lik.value, g if want_grad else None, rc)

    def query_batch(self, Xq, want_mu=True, want_var=True):
        Xq = _c(Xq)
        M = Xq.shape[0]
        assert Xq.shape[1] == self.D
        kta = np.zeros((M, self.P), order="F") if want_mu else None
        var = np.zeros(M) if want_var else None
        self._chk(self.lib.fn("query_batch")(self._h, _d(Xq), M, _d(kta) if want_mu else None,
                                             _d(var) if want_var else None), "query_batch")
        return kta, var

    # -- accessors
    def nb_samples(self) -> int:
        n = _i64()
        self._chk(self.lib.fn("nb_samples")(self._h, C.byref(n)), "nb_samples")
        return n.value

    def _get_mat(self, name):
        n = self.N
        A = np.zeros((n, n), order="F")
        self._chk(self.lib.fn(name)(self._h, _d(A), n), name)
        return A

    def get_L(self):
        return self._get_mat("get_L")

    def get_K(self):
        return self._get_mat("get_K")

    def get_Kinv(self):
        return self._get_mat("get_Kinv")

    def set_L(self, L):
        L = _c(L, "F")
        self._chk(self.lib.fn("set_L")(self._h, _d(L), L.shape[0]), "set_L")

    def get_alpha(self):
        a = np.zeros((self.N, self.P), order="F")
        self._chk(self.lib.fn("get_alpha")(self._h, _d(a)), "get_alpha")
        return a

    def set_alpha(self, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        a = _c(a, "F")
        self._chk(self.lib.fn("set_alpha")(self._h, _d(a)), "set_alpha")

    def synchronize(self):
        self._chk(self.lib.fn("synchronize")(self._h), "synchronize")

    # -- instrumentation (HIP library only)
    def get_stream(self) -> int:
        s = _vp()
        self._chk(self.lib.fn("get_stream")(self._h, C.byref(s)), "get_stream")
        return s.value or 0

    def set_profiling(self, on: bool):
        self._chk(self.lib.fn("set_profiling")(self._h, int(on)), "set_profiling")

    def reset_phase_ms(self):
        self._chk(self.lib.fn("reset_phase_ms")(self._h), "reset_phase_ms")

    def get_phase_ms(self):
        ms = np.zeros(PH_COUNT)
        fl = np.zeros(PH_COUNT)
        ln = (C.c_int64 * PH_COUNT)()
        self._chk(self.lib.fn("get_phase_ms")(self._h, _d(ms), ln, _d(fl), PH_COUNT), "get_phase_ms")
        return {PHASE_NAMES[i]: {"ms": float(ms[i]), "launches": int(ln[i]), "flops": float(fl[i])}
                for i in range(PH_COUNT)}


def sparsify(lib, X, max_points, device_id=0):
    """SparsifiedGP::_sparsify (sparsified_gp.hpp:157-183): indices of the samples that survive."""
    X = _c(X)
    N, D = X.shape
    keep = np.zeros(N, dtype=np.int64)
    n = _i64(0)
    rc = lib.fn("sparsify")(int(device_id), _d(X), N, D, int(max_points), keep.ctypes.data_as(C.POINTER(_i64)),
                            C.byref(n))
    if rc != 0:
        raise EngineError(f"sparsify failed with status {rc}")
    return keep[: n.value].copy()


def batch_compute(handles):
    lib = handles[0].lib
    arr = (_vp * len(handles))(*[h._h for h in handles])
    st = (C.c_int * len(handles))()
    rc = lib.fn("batch_compute")(arr, len(handles), st)
    if rc < 0:
        raise EngineError(f"batch_compute failed: {rc}")
    return list(st)


def batch_log_lik(handles):
    lib = handles[0].lib
    arr = (_vp * len(handles))(*[h._h for h in handles])
    out = np.zeros(len(handles))
    rc = lib.fn("batch_log_lik")(arr, len(handles), _d(out))
    if rc < 0:
        raise EngineError(f"batch_log_lik failed: {rc}")
    return out


def batch_hp_objective(handles, kind, thetas, noises, optimize_noise=False, want_grad=True):
    """gpe_batch_hp_objective: thetas (G x n_theta), noises (G,) -> (lik (G,), grad (G x n_grad) or None, status list)."""
    lib = handles[0].lib
    G = len(handles)
    kind = KERNEL_NAMES.get(kind, kind)
    th = _c(np.asarray(thetas, dtype=np.float64).reshape(G, -1))
    nz = _c(np.broadcast_to(np.asarray(noises, dtype=np.float64), (G,)).copy())
    n_theta = th.shape[1]
    n_grad = n_theta + (1 if optimize_noise else 0)
    arr = (_vp * G)(*[h._h for h in handles])
    lik = np.zeros(G)
    grad = np.zeros((G, n_grad))
    st = (C.c_int * G)()
    rc = lib.fn("batch_hp_objective")(arr, G, int(kind), _d(th), n_theta, _d(nz), int(optimize_noise), int(want_grad), _d(lik),
                                      _d(grad) if want_grad else None, st)
    if rc < 0:
        raise EngineError(f"batch_hp_objective failed: {rc}")
    for h in handles:
        h.n_theta = n_theta
        h.kind = kind
    return lik, (grad if want_grad else None), list(st)


def device_count(lib) -> int:
    n = C.c_int()
    rc = lib.fn("device_count")(C.byref(n))
    if rc < 0:
        raise EngineError(f"device_count failed: {rc}")
    return n.value
