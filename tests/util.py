"""Shared helpers for the parity tests (oracle and HIP engine are driven identically)."""
from pathlib import Path

import numpy as np

from limbo_amd import _capi

GOLDEN = Path(__file__).resolve().parent / "golden"


def golden_files(prefix):
    return sorted(GOLDEN.glob(prefix + "*.npz"))


def load(path):
    z = np.load(path, allow_pickle=False)
    return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}


class MemoLib:
    """The CPU oracle behind a disk memo (test infrastructure for test infrastructure): test_gpu_process_wide_switches runs the
    same parity script in 13 child processes that differ only in the ENGINE's environment; the oracle's answers (minutes of
    single-core O(N^3) work in all) are the same every time.  A handle made here digests its call sequence (method names and
    pickled arguments); a result the memo holds for that digest is returned, otherwise the real oracle handle is created, the
    calls so far are replayed, and it carries on for real.  What is compared with the engine is exactly what the oracle
    computed — once instead of 13 times."""

    def __init__(self, lib, path):
        import pickle

        self.lib, self.path, self.dirty = lib, Path(path), False
        self.store = pickle.loads(self.path.read_bytes()) if self.path.exists() else {}

    def handle(self):
        return _MemoHandle(self)

    def save(self):
        import os
        import pickle

        if self.dirty:
            tmp = self.path.with_suffix(".tmp%d" % os.getpid())
            tmp.write_bytes(pickle.dumps(self.store, protocol=4))
            tmp.replace(self.path)
            self.dirty = False


class _MemoHandle:
    def __init__(self, memo):
        import hashlib

        self._memo, self._log, self._real, self._digest = memo, [], None, hashlib.sha1(b"oracle handle")

    def _call(self, name, args, kwargs):
        import pickle

        d = self._digest.copy()
        d.update(name.encode())
        d.update(pickle.dumps((args, sorted(kwargs.items())), protocol=4))
        self._digest, key = d, d.hexdigest()
        self._log.append((name, args, kwargs))
        if self._real is None and key in self._memo.store:
            return self._memo.store[key]
        if self._real is None:
            self._real = _capi.Handle(self._memo.lib)
            for n_, a_, k_ in self._log[:-1]:
                getattr(self._real, n_)(*a_, **k_)
        out = getattr(self._real, name)(*args, **kwargs)
        self._memo.store[key] = out
        self._memo.dirty = True
        return out

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *a, **k: self._call(name, a, k)

    def close(self):
        if self._real is not None:
            self._real.close()
            self._real = None


def new_gp(lib, kind, X, obs_mean, theta, noise):
    h = lib.handle() if isinstance(lib, MemoLib) else _capi.Handle(lib)
    h.set_data(X, obs_mean)
    h.set_kernel(int(kind), theta, float(noise))
    return h


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def relerr_norm(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
