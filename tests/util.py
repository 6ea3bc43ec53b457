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


def new_gp(lib, kind, X, obs_mean, theta, noise):
    h = _capi.Handle(lib)
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
