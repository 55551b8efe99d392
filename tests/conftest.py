import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
# the library's test-only switches (PFD_EXACT_LEVELS, PFD_TEST_HCAP, ... forcing a fallback engine or shrinking a
# capacity) are inert unless this is set before the library is first used
os.environ.setdefault("PFD_ENABLE_KNOBS", "1")
# the kernels that take a workgroup per 256 chains of the exact-order engine (k_xtrunk_prescan, k_xtrunk_dscan_lds) are used
# for rounds of >= 2^20 chains; the test rasters are far smaller, so the suite lowers the threshold and runs EVERYTHING
# through them (test_gpu_large.py::test_exact_engine_accuflux and the 30000^2 test of test_gpu_fullsize.py run the production
# threshold: there the rounds lie on both sides of it)
os.environ.setdefault("PFD_TEST_FUSE_MIN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O

    O.build()
    return O


def case_names():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        m = json.load(f)
    return sorted(k for k in m if not k.startswith("_"))


@pytest.fixture(scope="session")
def gpu_lib():
    """libpfd_hip with a visible device; GPU tests fail (not skip) without it."""
    from pyflwdir_amd import _hip

    lib = _hip.lib()
    assert _hip.device_count() >= 1, "no HIP device visible: -m gpu tests need an MI355X"
    return lib
