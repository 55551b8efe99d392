"""rank and the exact core.idxs_seq order with 64-bit cell indices (csrc/order64.hip) — the int64 rung of the reference's
index ladder (pyflwdir/pyflwdir.py:105-127), for rasters beyond 2**32 - 2 cells.  The form needs no level engine, so it
runs on ANY raster; PFD_TEST_ORDER64 (with PFD_ENABLE_KNOBS=1) makes the small golden cases take it, against the
reference's own outputs bit for bit (core.idxs_seq, pyflwdir/core.py:87-117; core.rank, core.py:17-47), and a seeded raster
against the 32-bit form and the oracle.  The at-size run is tools/big_frontend_probe.py (profiles/r05_big_frontend.txt)."""
import os
import time

import numpy as np
import pytest

from conftest import case_names
from golden_util import Case

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wide(gpu_lib):
    os.environ["PFD_TEST_ORDER64"] = "1"  # (conftest.py enables the knobs)
    try:
        yield
    finally:
        for k in ("PFD_TEST_ORDER64", "PFD_TEST_ORDER64_SMALL"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("small", [None, "48"])
@pytest.mark.parametrize("name", case_names())
def test_golden_cases_through_the_64_bit_form(name, small, manifest, wide):
    """Both level forms: one workgroup per level (default up to 16384 cells) and count / scan / scatter (threshold 48)."""
    import pyflwdir_amd as pyflwdir

    case = Case(name, manifest)
    st = case.entry["stats"]
    if small:
        os.environ["PFD_TEST_ORDER64_SMALL"] = small
    flw = pyflwdir.from_array(case.d8, ftype="d8", cache=False)
    assert flw._wide()
    if st["n_loop_cells"]:
        # the reference marks the cells of a cycle (and everything draining to it) with rank -1 and leaves them out of the
        # sequence (core.py:17-47, :87-117); the 64-bit form walks from the pits like it does
        assert not flw.isvalid
        case.check("rank", flw.rank.ravel())
        assert int(np.count_nonzero(flw.rank == -1)) == st["n_loop_cells"]
        case.check("idxs_seq_int32", flw.idxs_seq)
        assert flw.idxs_seq.size == st["n_seq"] == st["n_valid"] - st["n_loop_cells"]
        assert flw.nnodes == st["n_seq"]
        return
    assert flw.isvalid
    case.check("rank", flw.rank.ravel())
    case.check("idxs_seq_int32", flw.idxs_seq)
    if "idxs_seq_int64" in case.digests:
        case.check("idxs_seq_int64", flw._h.idxs_seq(np.int64))
    assert flw.nnodes == st["n_seq"]
    # "sort" is the reference's numpy expression over the 64-bit form's ranks: rank-monotone, the property the
    # reference tests (tests/test_core.py:82)
    flw.order_cells(method="sort")
    assert flw.idxs_seq.size == st["n_seq"] and np.all(np.diff(flw.rank.ravel()[flw.idxs_seq]) >= 0)


def test_basins_and_ucat_area_of_a_cyclic_raster_beyond_32_bits(manifest, wide, monkeypatch):
    """The one-handle label query is only taken on acyclic rasters (an outlet downstream of a cycle would hide it), and so
    is the row-block protocol (pfd_basins_begin): with cycles AND 64-bit cells the library answers PFD_EUNSUPPORTED, the
    front end passes it on — a stated limit (the 32-bit object answers through the level engine)."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    case = Case("synth_loops_96x80", manifest)
    tf = Affine(0.01, 0, 4.0, 0, -0.01, 52.0)
    monkeypatch.delenv("PFD_TEST_ORDER64")
    whole = pyflwdir.from_array(case.d8, ftype="d8", transform=tf, latlon=True, cache=False)
    valid = np.flatnonzero(case.d8.ravel() != 247)
    outs = np.random.default_rng(1).choice(valid, 24, replace=False).astype(whole.idxs_ds.dtype)
    exp_b = whole.basins(idxs=outs)
    exp_u = {u: whole.ucat_area(outs, unit=u) for u in ("cell", "km2")}
    monkeypatch.setenv("PFD_TEST_ORDER64", "1")
    monkeypatch.setenv("PFD_TEST_BIG_CELLS", "2000")
    big = pyflwdir.from_array(case.d8, ftype="d8", transform=tf, latlon=True, cache=False)
    assert big._wide() and big._row_blocks_needed() > 1
    with pytest.raises(NotImplementedError):
        big._h.basins(outs.astype(np.int64), np.arange(1, 25, dtype=np.uint32))
    assert exp_b.shape == case.shape and exp_u["km2"][1].dtype == np.float64
    with pytest.raises(NotImplementedError, match="cycles"):
        big.basins(idxs=outs)
    for u in ("cell", "km2"):
        with pytest.raises(NotImplementedError, match="cycles"):
            big.ucat_area(outs, unit=u)


@pytest.mark.parametrize("shape,small", [((700, 900), None), ((1500, 1100), "1000"), ((129, 4097), "0")])
def test_seeded_raster_against_the_32_bit_form_and_the_oracle(shape, small, gpu_lib, oracle, wide):
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(shape[0], shape[1], seed=11, tilt=3000, white=3, nodata_pct=7)
    if small is not None:
        os.environ["PFD_TEST_ORDER64_SMALL"] = small
    h = _hip.RasterHandle(d8, shape[0], shape[1])
    seq64 = h.idxs_seq(np.int64)
    rank64 = h.rank()
    os.environ.pop("PFD_TEST_ORDER64")
    h2 = _hip.RasterHandle(d8, shape[0], shape[1])
    assert not h2.wide_cells()
    np.testing.assert_array_equal(seq64, h2.idxs_seq(np.int64))
    np.testing.assert_array_equal(rank64, h2.rank())
    idxs_ds, idxs_pit, _ = oracle.from_array(d8, dtype=np.int64)
    np.testing.assert_array_equal(seq64, oracle.idxs_seq(idxs_ds, idxs_pit))
    h.close()
    h2.close()


def test_int64_is_the_only_index_dtype_of_the_c_entry(gpu_lib, oracle, wide):
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(200, 300, seed=1)
    h = _hip.RasterHandle(d8, 200, 300)
    out = np.empty(d8.size, np.int32)
    rc = _hip.lib().pfd_idxs_seq(h._h, _hip.PFD_I32, _hip.ptr(out), _hip.PFD_HOST)
    assert rc == -1 and b"int64" in _hip.lib().pfd_last_error()  # PFD_EINVAL
    h.close()


def test_true_size_beyond_2_32_cells(gpu_lib):
    """66000 x 66000 = 4.356e9 cells (> 2**32 - 2: the int64 rung of pyflwdir.py:105-127) through the front end at TRUE size —
    no lowered threshold: the order-free fixed-point `upstream_area("km2")` on one handle against the exact row-block form,
    rank and the exact idxs_seq order by the properties that define them (core.py:17-47, :87-117, tests/test_core.py:66-82),
    the classic stream order over row blocks by its local properties (streams.py:191-225), basins and ucat_area on one handle
    (labels follow the downstream cell; counts and float64 areas against a row-count dot product), floodplains over streamed
    row blocks by its local equations (dem.py:333-379), and the same raster with a cycle
    injected: the cells that never reach a pit read -1 and are left out of the sequence.  The body is
    tests/true_size_child.py, in a child process (it holds a 120 GiB arena)."""
    import subprocess
    import sys

    import gc

    from pyflwdir_amd import _hip

    # the pytest process gives back what it holds on the GPU (the default arena and the allocator's idle blocks of the tests
    # before this one): the child needs most of the HBM
    gc.collect()
    _hip.reserve(0)
    _hip.check(_hip.lib().pfd_trim(0))
    env = {k: v for k, v in os.environ.items() if k not in ("PFD_TEST_ORDER64", "PFD_TEST_ORDER64_SMALL", "PFD_TEST_BIG_CELLS")}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "true_size_child.py")],
                         capture_output=True, text=True, timeout=2400, env=env)
    assert out.returncode == 0 and "true size: ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    print(out.stdout[-600:])
