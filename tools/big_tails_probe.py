"""The cell-local exports of the front end beyond 2**32 - 2 cells — row slices assembled on the host (raster.py `_sliced`):
idxs_ds, main_upstream, n_upstream — timed at SIZE x SIZE (default 90000 = 8.1 Gcells), with a sampled check against their
definitions.  VERDICT r05 item 6: main_upstream 102 s, idxs_ds 30 s before the copies ran in host threads.

    python tools/big_tails_probe.py [SIZE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip
size = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
_hip.reserve(100 << 30)
buf = _hip.synth_d8_device(size, size, seed=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
print(f"{size}x{size} = {d8.size / 1e9:.2f} Gcells on the host", flush=True)
t0 = time.perf_counter(); flw = pyflwdir.from_array(d8, ftype="d8", cache=False); print(f"  from_array {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter(); upa = flw.upstream_area(); print(f"  upstream_area {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter(); ds = flw.idxs_ds; t_ds = time.perf_counter() - t0
dr = np.array([0, 1, 1, 1, 0, -1, -1, -1]); dc = np.array([1, 1, 0, -1, -1, -1, 0, 1])
ok = True
for r0 in (0, size // 2 + 17, size - 1000):  # idxs_ds of a direction code = the neighbour it names; a pit names itself
    blk = d8[r0:r0 + 1000]
    base = (np.arange(r0, r0 + 1000, dtype=np.int64)[:, None] * size + np.arange(size, dtype=np.int64)[None, :])
    got = ds[r0 * size:(r0 + 1000) * size].reshape(1000, size)
    for k in range(8):
        m = blk == (1 << k)
        m[0] = m[-1] = False; m[:, 0] = m[:, -1] = False
        ok = ok and bool(np.all(got[m] == base[m] + dr[k] * size + dc[k]))
    ok = ok and bool(np.all(got[blk == 0] == base[blk == 0]))
print(f"  idxs_ds {t_ds:.1f} s  dtype {ds.dtype}  sampled rows equal their definition: {ok}", flush=True)
del ds
t0 = time.perf_counter(); mu = flw.main_upstream(upa); t_mu = time.perf_counter() - t0
ok = True
for r0 in (1, size // 2 + 17, size - 1001):  # the main upstream cell drains into the cell and no upstream cell has a larger area
    got = mu[r0 * size:(r0 + 1000) * size]
    idx = np.flatnonzero(got >= 0)[:: 997]
    x = idx + r0 * size
    ok = ok and bool(np.all(flw.idxs_ds[:0].dtype == np.int64)) if False else ok
    src = got[idx]
    rr, cc = np.divmod(src, size)
    code = d8[rr, cc]
    k = np.log2(code).astype(int)
    ok = ok and bool(np.all((rr + dr[k]) * size + cc + dc[k] == x))
print(f"  main_upstream {t_mu:.1f} s  dtype {mu.dtype}  sampled main upstream cells drain into their cell: {ok}", flush=True)
del mu
t0 = time.perf_counter(); nu = flw.n_upstream; print(f"  n_upstream {time.perf_counter() - t0:.1f} s  histogram of 1e7 cells {np.bincount(nu.ravel()[:10_000_000] + 9)[9:14].tolist()}", flush=True)
del nu
# basins and ucat_area on ONE handle (round 6: the tiled label query runs at any size; the float sums of ucat_area walk the 64-bit
# sequence on the device in pieces, csrc/subgrid.hip) — on a lat/lon grid (float64 areas) against the host-composed form
from pyflwdir_amd._affine import Affine
from pyflwdir_amd import gis
res = 1.0 / 1200.0
flw = pyflwdir.from_array(d8, ftype="d8", transform=Affine(res, 0.0, 5.0, 0.0, -res, 80.0), latlon=True, cache=False)
top = np.argsort(upa.ravel()[:: 4097])[-500:].astype(np.int64) * 4097
del upa
t0 = time.perf_counter(); bas = flw.basins(idxs=top); t_b = time.perf_counter() - t0
print(f"  basins, 500 outlets {t_b:.1f} s  dtype {bas.dtype}  labelled cells {int(np.count_nonzero(bas[::7]))} of {bas[::7].size} sampled", flush=True)
t0 = time.perf_counter(); m_c, a_c = flw.ucat_area(top, unit="cell"); t_c = time.perf_counter() - t0
ok_c = bool(np.array_equal(m_c[::5], bas[::5]))
print(f"  ucat_area cell, 500 outlets {t_c:.1f} s  map dtype {m_c.dtype}  equals basins (sampled): {ok_c}  largest {np.sort(a_c)[-2:].tolist()}", flush=True)
del m_c, bas
t0 = time.perf_counter(); m_k, a_k = flw.ucat_area(top, unit="km2"); t_k = time.perf_counter() - t0
print(f"  ucat_area km2, 500 outlets {t_k:.1f} s  area dtype {a_k.dtype}  largest {np.sort(a_k)[-2:].tolist()}", flush=True)
del m_k
if os.environ.get("PFD_PROBE_COMPOSED", "1") == "1":
    rows = np.ascontiguousarray(gis.area_rows(flw.transform, flw.shape, flw.latlon, unit="m2") / gis.AREA_FACTORS["km2"])
    t0 = time.perf_counter(); m2, a2 = flw._ucat_area_wide(top.copy(), rows); t2 = time.perf_counter() - t0
    print(f"  the host-composed form (row-block basins + np.add.at over the sequence): {t2:.1f} s  areas bit-identical: {a2.tobytes() == a_k.tobytes()}", flush=True)
