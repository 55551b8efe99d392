import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from pyflwdir_amd import dist, _hip
import pyflwdir_amd as pf
for shape,seed,kw in [((700,900),11,dict(tilt=1<<26,white=2,nodata_pct=0)), ((1500,2100),3,dict(tilt=1<<26,white=2,nodata_pct=0)), ((300,400),9,dict(tilt=1<<26,white=2,nodata_pct=0))]:
    d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
    exp,_,_ = O.upstream_area_cell(d8)
    h=_hip.RasterHandle(d8, shape[0], shape[1]); h.set_profiling(True)
    got = h.upstream_area_cell().reshape(shape)
    print(shape, 'single handle equal:', np.array_equal(got,exp), [s['name'] for s in h.last_timing()])
    bad = np.argwhere(got!=exp)
    if bad.size:
        print(' nbad', len(bad), 'first', bad[:5], got[tuple(bad[0])], exp[tuple(bad[0])], 'rows', np.unique(bad[:,0]//64)[:10], 'cols', np.unique(bad[:,1]//64)[:10])
    g2 = dist.upstream_area_blocks(d8, 1)
    print('  blocks(1) equal:', np.array_equal(g2, exp))
