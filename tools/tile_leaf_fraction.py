"""What a rake / leaf-peel of the tile passes could remove: the share of cells of a 64 x 64 tile that no cell of the same tile
drains into (in-tile leaves), that only leaves drain into (second level), and the in-degree mix — on a slice of the bench
raster (rows of the 90000 x 90000 synthetic raster, generated on the device, counted with numpy).

    python tools/tile_leaf_fraction.py [SIZE] [ROWS] [REGIME]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyflwdir_amd import _hip

size = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
regimes = {"river": dict(seed=0), "rough": dict(seed=0, tilt=100000, white=2), "meander": dict(seed=0, tilt=3000, white=2)}
DR = {1: 0, 2: 1, 4: 1, 8: 1, 16: 0, 32: -1, 64: -1, 128: -1}
DC = {1: 1, 2: 1, 4: 0, 8: -1, 16: -1, 32: -1, 64: 0, 128: 1}
for name in ([sys.argv[3]] if len(sys.argv) > 3 else list(regimes)):
    row0 = (size // 2) // 64 * 64
    buf = _hip.synth_d8_device(size, size, row0=row0, nrows=rows, **regimes[name])
    ncol = size // 64 * 64
    d8 = buf.download(np.uint8, (rows, size))[:, :ncol]
    buf.free()
    r, c = np.indices(d8.shape, dtype=np.int32)
    tr, tc = np.full(d8.shape, -1, np.int32), np.full(d8.shape, -1, np.int32)
    for k in DR:
        m = d8 == k
        tr[m] = r[m] + DR[k]
        tc[m] = c[m] + DC[k]
    same = (tr >= 0) & (tr < rows) & (tc >= 0) & (tc < ncol)
    same &= ((tr >> 6) == (r >> 6)) & ((tc >> 6) == (c >> 6))
    flat = tr.astype(np.int64) * ncol + tc
    indeg = np.bincount(flat[same], minlength=d8.size).reshape(d8.shape)
    leaf = indeg == 0
    nonleaf_kids = np.bincount(flat[same & ~leaf], minlength=d8.size).reshape(d8.shape)
    lvl2 = ~leaf & (nonleaf_kids == 0)
    print(f"{name:8s} rows {row0}..{row0 + rows} of {size}^2: in-tile leaves {leaf.mean():.3f}, cells only leaves drain into "
          f"{lvl2.mean():.3f}, in-degree 1 / >= 2: {np.mean(indeg == 1):.3f} / {np.mean(indeg >= 2):.3f}", flush=True)
