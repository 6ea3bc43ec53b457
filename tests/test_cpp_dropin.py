"""The C++ drop-in `limbo::model::GP` (include/limbo_amd/limbo/model/gp.hpp) against the
reference's own test cases (src/tests/test_gp.cpp re-expressed without Boost, tests/cpp/).
CPU: the header set compiles with the host compiler and links against libgpengine.so.
GPU: the 15 cases run on the device (exit code = failed cases)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CPP = ROOT / "tests" / "cpp"


def _build():
    subprocess.check_call(["make", "-C", str(CPP)], stdout=subprocess.DEVNULL)
    exe = CPP / "test_gp_dropin"
    assert exe.exists()
    return exe


def test_dropin_header_compiles_and_links():
    from limbo_amd import _capi

    assert _capi.ENGINE_SO.exists(), "libgpengine.so not built: run __graft_entry__.build()"
    _build()


def test_dropin_keeps_the_policy_api_names():
    """template parameters, members and protected names the reference's callers rely on"""
    src = (ROOT / "include" / "limbo_amd" / "limbo" / "model" / "gp.hpp").read_text()
    for name in ("void compute(", "void add_sample(", "query(const Eigen::VectorXd& v) const", "Eigen::VectorXd mu(",
                 "double sigma(", "void optimize_hyperparams()", "void recompute(bool update_obs_mean = true, bool update_full_kernel = true)",
                 "double compute_log_lik()", "compute_kernel_grad_log_lik()", "compute_mean_grad_log_lik()",
                 "void compute_inv_kernel()", "bool inv_kernel_computed()", "matrixL() const", "alpha() const",
                 "_samples;", "_observations;", "_kernel_function;", "_mean_function;", "_matrixL;", "_alpha;",
                 "using GPBasic", "using GPOpt"):
        assert name in src, name


@pytest.mark.gpu
@pytest.mark.parametrize("virtual_devices,min_n", [(0, None), (4, None), (0, 0)])
def test_dropin_cases_on_device(virtual_devices, min_n):
    """all cases on the visible device(s); then once more with 4 logical devices dealt over the physical ones
    (GPE_VIRTUAL_DEVICES): MultiGP members, restart clones and explicit copies land on devices 0..3 through
    gpe_clone_to — the multi-GPU placement of the drop-in, exercised on a one-GPU box.  Round 4: by default models below
    Params::gpu::min_n_for_gpu (256) samples live on the host (most of these cases' models are that small: they exercise
    the host path and its hand-over to the device); LIMBO_AMD_MIN_N_FOR_GPU=0 runs everything on the device as rounds 1-3 did."""
    import os

    exe = _build()
    env = dict(os.environ)
    if min_n is not None:
        env["LIMBO_AMD_MIN_N_FOR_GPU"] = str(min_n)
    if virtual_devices:
        env["GPE_VIRTUAL_DEVICES"] = str(virtual_devices)
        env["LIMBO_AMD_HOST_BATCH_CROSSOVER"] = "64"  # host-resident models answer query_batch from their device shadow much earlier
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failed cases" in r.stdout
    if virtual_devices:
        assert f"{virtual_devices} visible device(s)" in r.stdout


@pytest.mark.gpu
def test_dropin_with_limbos_own_tree_behind_it():
    """INTEGRATION.md's include-path switch with /root/reference/src behind the drop-in: limbo's opt::GridSearch /
    RandomPoint / tools over the device model in a boptimizer.hpp-shaped loop (tests/cpp/test_mixed_tree.cpp).
    The binary is built where the reference sources are and travels with the snapshot."""
    _build()
    exe = CPP / "test_mixed_tree"
    if not exe.exists():
        pytest.skip("test_mixed_tree was not built (no /root/reference at build time)")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
