import sys
sys.path.insert(0, '.')
import numpy as np
from pyflwdir_amd import _hip, dist
for (nrow, ncol, kw) in [(1, 5_000_000, dict(tilt=100000)), (4_000_000, 1, dict(tilt=100000)), (3, 20_000_000, dict(tilt=100000, nodata_pct=20)),
                         (200_000, 7, dict(tilt=1 << 26)), (65, 1_000_003, dict(tilt=1 << 26, nodata_pct=10))]:
    d8 = _hip.synth_d8_device(nrow, ncol, seed=5, white=2, **kw)
    h = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE)
    tiled = h.upstream_area_cell()
    levels = h.upstream_area_cell(engine="levels")
    hd = _hip.RasterHandle(d8, nrow, ncol, device=0, memspace=_hip.PFD_DEVICE, deferred=True)
    deferred = hd.upstream_area_cell()
    codes = d8.download(np.uint8, (nrow, ncol))
    ok_blocks = True
    if nrow >= 8:
        ok_blocks = np.array_equal(dist.upstream_area_blocks(codes, 3).ravel(), tiled)
    info = h.info()
    pits = h.idxs_pit(np.int32)
    print((nrow, ncol), "tiled==levels", np.array_equal(tiled, levels), "deferred", np.array_equal(tiled, deferred), "blocks", ok_blocks,
          "pit-sum", int(tiled[pits].astype(np.int64).sum()) == info["n_valid"], "levels", h.info()["n_levels"], flush=True)
    h.close(); hd.close()
