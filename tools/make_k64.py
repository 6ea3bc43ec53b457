#!/usr/bin/env python
"""tools/tmp/K64.bin: the first 64 x 64 diagonal block of the headline kernel matrix (configs[1]: SE-ARD, D=6, unit
length scales, noise 0.01), row-major float64 — the test matrix of tools/diagbench.hip / tools/diagflow."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from limbo_amd import synth  # noqa: E402

X, _ = synth.make_problem("c2", N=64)
d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
K = np.exp(-0.5 * d2) + 0.01 * np.eye(64)
out = Path(__file__).resolve().parent / "tmp"
out.mkdir(exist_ok=True)
K.astype(np.float64).tofile(out / "K64.bin")
print(f"wrote {out / 'K64.bin'}  (cond {np.linalg.cond(K):.0f})")
