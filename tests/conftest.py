"""Shared test plumbing.

`-m "not gpu"`: oracle vs golden vectors, host mirror, C-ABI surface (no compute
on a GPU).  `-m gpu`: the parity tests proper, all through the C ABI of
libcurdlemsm.so.  Nothing here reads /root/reference.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "go-curdleproofs_amd")
for p in (os.path.join(ROOT, "oracle", "py"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(PKG, "libcurdlemsm.so")
    ora = os.path.join(ROOT, "oracle", "libcurdle_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", PKG, "-j4"])
    if not os.path.exists(ora) or not os.path.exists(os.path.join(ROOT, "oracle", "libcurdle_cpufast.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


_ensure_built()


@pytest.fixture(scope="session")
def cm():
    import curdlemsm
    return curdlemsm


@pytest.fixture(scope="session")
def oracle():
    import bls12381_ref
    bls12381_ref.self_check()
    return bls12381_ref


@pytest.fixture(scope="session")
def coracle():
    import coracle as co
    return co


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "msm_vectors.npz"))


@pytest.fixture(scope="session")
def golden_acc():
    return np.load(os.path.join(ROOT, "tests", "golden", "accumulator_vectors.npz"))


@pytest.fixture(scope="session")
def gpu(cm):
    """The HIP path must be the one that runs: no device -> hard failure, not a skip to a fallback."""
    if not cm.device_available():
        pytest.fail("gpu-marked test run without a visible HIP device")
    cm.init(0)
    return cm


def golden_case_names(npz):
    return sorted(k[: -len("_expected")] for k in npz.files if k.endswith("_expected"))
