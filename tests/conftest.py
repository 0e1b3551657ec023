"""pytest configuration: `gpu` marker, shared paths and fixtures.

`python -m pytest tests -m "not gpu"` runs here (no GPU): the oracle against the
reference-derived golden fixtures and the reference's own known-answer tests,
the host C logic of libcmusphinx_amd through its C ABI, and the ABI itself.
`python -m pytest tests -m gpu` runs on the MI355X box: the HIP path through
the C ABI against the oracle and the same fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def tidigits_dir():
    return os.path.join(GOLDEN, "tidigits")


@pytest.fixture(scope="session")
def olm():
    import oracle_lib as O
    return O.OracleLogMath(1.0003, 0, 1)


@pytest.fixture
def variants(gpu_lib):
    """force kernel variants for one test (s3a_set_variants is process-wide): restored to the defaults afterwards"""
    yield gpu_lib.set_variants
    gpu_lib.set_variants()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a box with a GPU.  Never falls back."""
    from cmusphinx_amd import lib
    lib.load()
    if lib.device_count() < 1:
        pytest.fail("gpu-marked test running without a usable HIP device")
    return lib
