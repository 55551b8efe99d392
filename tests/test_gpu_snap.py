"""core.snap behind FlwdirRaster.snap / basins(streams=...) / add_pits(streams=...) against the reference's
recorded outputs (tests/golden/wide_snap.npz, oracle/gen_golden_wide.py)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["flwdir0", "flwdir_large", "synth_rough_nodata_384x512", "rhine"])
def test_snap_and_streams(gpu_lib, name):
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    W = np.load(os.path.join(GOLD, "wide_snap.npz"))
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ent = json.load(open(os.path.join(GOLD, "manifest.json")))[name]
    flw = pyflwdir.from_array(z["d8"], ftype="d8", transform=Affine(*ent["transform"]), latlon=ent["latlon"], cache=False)
    idxs, streams = W[f"in_{name}_idxs"], W[f"in_{name}_streams"]
    for key, kw in (("snap", dict(mask=streams)), ("snap5", dict(mask=streams, max_length=5)), ("snap_nomask", {})):
        i, d = flw.snap(idxs=idxs, **kw)
        ei, ed = W[f"out_{name}_{key}_idxs"], W[f"out_{name}_{key}_dist"]
        assert i.dtype == ei.dtype and d.dtype == ed.dtype
        assert np.array_equal(i, ei) and np.array_equal(d, ed), key
    assert np.array_equal(flw.basins(idxs=idxs[:40], streams=streams), W[f"out_{name}_basins_streams"])
    flw2 = pyflwdir.from_array(z["d8"], ftype="d8", cache=False)
    flw2.add_pits(idxs=idxs[:10], streams=streams)
    assert np.array_equal(flw2.idxs_pit, W[f"out_{name}_addpits_streams_idxs_pit"])
    assert np.array_equal(flw2.upstream_area(), W[f"out_{name}_addpits_streams_upa"])
    # add_pits is atomic: a bad index changes nothing
    before = flw2.idxs_pit.copy()
    with pytest.raises(IndexError):
        flw2.add_pits(idxs=np.array([3, flw2.size + 5]))
    assert np.array_equal(flw2.idxs_pit, before)


@pytest.mark.parametrize("name", ["flwdir0", "flwdir_large", "synth_rough_nodata_384x512", "rhine"])
def test_snap_upstream_and_metres(gpu_lib, name):
    """Flwdir.snap upstream (along the main upstream cells) and in metres (reference pyflwdir/flwdir.py:500-560,
    core.py:308-366,440-480) against the reference's recorded outputs (tests/golden/wide_snap2.npz)."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd._affine import Affine

    W = np.load(os.path.join(GOLD, "wide_snap.npz"))
    W2 = np.load(os.path.join(GOLD, "wide_snap2.npz"))
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ent = json.load(open(os.path.join(GOLD, "manifest.json")))[name]
    flw = pyflwdir.from_array(z["d8"], ftype="d8", transform=Affine(*ent["transform"]), latlon=ent["latlon"], cache=False)
    idxs, streams, heads = W[f"in_{name}_idxs"], W[f"in_{name}_streams"], W2[f"in_{name}_heads"]
    cases = dict(down_m=dict(mask=streams, unit="m"), down_m_max=dict(mask=streams, unit="m", max_length=2500.0),
                 down_m_nomask=dict(unit="m"), up_cell=dict(direction="up"), up_cell_mask=dict(direction="up", mask=heads),
                 up_cell_max=dict(direction="up", max_length=7), up_m=dict(direction="up", unit="m"),
                 up_m_max=dict(direction="up", unit="m", mask=heads, max_length=4000.0))
    for key, kw in cases.items():
        i, d = flw.snap(idxs=idxs, **kw)
        ei, ed = W2[f"out_{name}_{key}_idxs"], W2[f"out_{name}_{key}_dist"]
        assert i.dtype == ei.dtype and d.dtype == ed.dtype, key
        assert np.array_equal(i, ei), key
        assert np.array_equal(d, ed), (key, np.flatnonzero(d != ed)[:5])
    with pytest.raises(ValueError, match="Unknown unit"):
        flw.snap(idxs=idxs, unit="km")
    with pytest.raises(ValueError, match="Unknown flow direction"):
        flw.snap(idxs=idxs, direction="sideways")


@pytest.mark.parametrize("name", ["flwdir0", "flwdir_large"])
def test_accuflux_narrow_integer_payloads(gpu_lib, name):
    """int8 / int16 / uint8 / uint16 payloads (the reference accumulates in the payload's own dtype, streams.py:36):
    identical where the sums stay in range, refused where the reference's would wrap."""
    import pyflwdir_amd as pyflwdir

    W2 = np.load(os.path.join(GOLD, "wide_snap2.npz"))
    z = np.load(os.path.join(GOLD, name + ".npz"))
    flw = pyflwdir.from_array(z["d8"], ftype="d8", cache=False)
    exact = 0
    for nm in ("int8", "int16", "uint8", "uint16"):
        data = W2[f"in_{name}_{nm}"]
        for key, kw in (("up", dict(nodata=2)), ("down", dict(nodata=-9999, direction="down"))):
            exp = W2[f"out_{name}_{nm}_{key}"]
            try:
                got = flw.accuflux(data, **kw)
            except NotImplementedError:
                # refused: then some accumulated magnitude really leaves the dtype's range
                wide = flw.accuflux(np.abs(data.astype(np.int64)), nodata=-9999, direction=kw.get("direction", "up"))
                assert wide.max() > min(np.iinfo(data.dtype).max, -int(np.iinfo(data.dtype).min) or np.iinfo(data.dtype).max)
                continue
            exact += 1
            assert got.dtype == exp.dtype and np.array_equal(got, exp), (nm, key)
    assert exact >= 3


def test_accuflux_int16_with_negative_nodata_is_not_refused(gpu_lib, oracle):
    """The usual int16 payload: small values, -9999 on missing cells.  The range check honours the nodata rule (a
    missing cell never enters a sum), so only a real overflow is refused; the result is the int32 accumulation."""
    import pyflwdir_amd as pyflwdir

    d8 = oracle.synth_d8(300, 260, seed=9, tilt=100000, white=2, nodata_pct=10).reshape(300, 260)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 3, d8.shape).astype(np.int16)
    data[rng.random(d8.shape) < 0.05] = -9999
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    for direction in ("up", "down"):
        got = flw.accuflux(data, nodata=-9999, direction=direction)
        exp = flw.accuflux(data.astype(np.int32), nodata=-9999, direction=direction)
        assert got.dtype == np.int16 and np.array_equal(got.astype(np.int32), exp)
    big = np.where(data == -9999, data, 30000).astype(np.int16)  # (sums beyond int16: the reference would wrap)
    with pytest.raises(NotImplementedError, match="leave the range"):
        flw.accuflux(big, nodata=-9999)
