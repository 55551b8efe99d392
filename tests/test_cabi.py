"""The C-ABI library loads and exports every symbol include/pfd.h declares (no compute calls:
this runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pfd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pfd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from pyflwdir_amd import _hip

    lib = _hip.lib()
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"libpfd_hip.so does not export {name}"
    # the binding's list is the header's list
    assert sorted(_hip.SYMBOLS) == names


def test_abi_version_and_error_string():
    from pyflwdir_amd import _hip

    lib = _hip.lib()
    assert lib.pfd_abi_version() == 1
    assert isinstance(lib.pfd_last_error(), bytes)


def test_no_silent_fallback_without_device():
    """On a box without a GPU the product path must raise, never fall back to a CPU path."""
    import numpy as np

    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    if _hip.device_count() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        pyflwdir.from_array(np.array([[1, 4], [0, 0]], dtype=np.uint8), ftype="d8")


def test_product_does_not_import_oracle():
    """Nothing under pyflwdir_amd/ may import or load the oracle."""
    pkg = os.path.join(ROOT, "pyflwdir_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
