"""Flwdir.upstream_sum on the GPU against the reference's own outputs (tests/golden/wide_arith.npz, recorded by
oracle/gen_golden_wide.py from /root/reference/pyflwdir/arithmetics.py:147-169) and its test-suite's identity
upstream_sum(ones) == n_upstream (tests/test_pyflwdir.py:191)."""
import ast
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_upstream_sum_golden(gpu_lib):
    import pyflwdir_amd as pyflwdir

    z = np.load(os.path.join(GOLD, "wide_arith.npz"))
    cases = ast.literal_eval(str(z["cases"]))
    assert len(cases) >= 16
    flws = {}
    for key in cases:
        name = key.rsplit("_", 1)[0]
        if name not in flws:
            d8 = np.load(os.path.join(GOLD, name + ".npz"))["d8"]
            flws[name] = pyflwdir.from_array(d8, ftype="d8", check_ftype=False, cache=False)
        data = z["in_" + key]
        got = flws[name].upstream_sum(data, mv=-9999)
        exp = z["out_" + key]
        assert got.dtype == exp.dtype and got.shape == exp.shape
        assert np.array_equal(got, exp), key
    for name, flw in flws.items():
        ones = np.ones(flw.shape, np.float64)
        got = flw.upstream_sum(ones, mv=np.nan)
        assert np.array_equal(got, z[f"out_{name}_ones_nan"]), name
        assert np.all(got.flat[flw.mask] == flw.n_upstream.flat[flw.mask])  # reference tests/test_pyflwdir.py:191
    with pytest.raises(ValueError, match="size does not match"):
        flws["flwdir0"].upstream_sum(np.ones((2, 1)))
