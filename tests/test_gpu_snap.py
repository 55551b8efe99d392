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
