#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generate the golden vectors under tests/golden/.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It imports the reference package itself (pure Python; executed
interpreted through the identity-``njit`` shim in oracle/refshim — the mode the
reference's own test-suite forces with NUMBA_DISABLE_JIT=1, reference
tests/conftest.py:7) and records its outputs for the hot path:

    core_d8.from_array / core.upstream_count / core.idxs_seq / core.rank and the
    FlwdirRaster methods upstream_area, accuflux (up & down), stream_order (Strahler and classic),
    basins, hand, main_upstream / idxs_us_main, stream_distance.

Inputs are the reference's own test rasters (tests/data/flwdir.asc, flwdir1.asc, the seeded
from_dem raster of tests/conftest.py:57-60, examples/rhine_d8.tif + rhine_elv0.tif) and
synthetic rasters from this repo's generator (oracle.synth_d8).  Small cases store full
output arrays; every case stores sha256 digests of the raw output bytes in manifest.json.
Nothing of the reference's source travels: fixtures are data only.

Usage:  python oracle/gen_golden.py [--only CASE]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != HERE]
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
os.environ["NUMBA_DISABLE_JIT"] = "1"

import pyflwdir  # noqa: E402  (the reference)
from pyflwdir import core, core_d8  # noqa: E402

from oracle import oracle as O  # noqa: E402  (only for the synthetic inputs)
from oracle import golden_inputs as GI  # noqa: E402
from pyflwdir_amd._affine import Affine  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FULL_LIMIT = 40000  # cases with at most this many cells store full output arrays


def digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    h.update(str(a.dtype.str).encode())
    h.update(str(a.shape).encode())
    h.update(a.tobytes())
    return h.hexdigest()


def read_tif(path):
    from PIL import Image

    im = Image.open(path)
    arr = np.array(im)
    tags = im.tag_v2
    sx, sy, _ = tags[33550]
    _, _, _, x0, y0, _ = tags[33922]
    return arr, (sx, 0.0, x0, 0.0, -sy, y0)


def inject_loops(d8):
    """Deliberately invalid raster: a 2-cycle and a 3-cycle (cells that never reach a pit)."""
    d8 = d8.copy()
    r, c = d8.shape[0] // 3, d8.shape[1] // 3
    d8[r, c], d8[r, c + 1] = 1, 16  # E <-> W
    r2, c2 = 2 * d8.shape[0] // 3, d8.shape[1] // 2
    d8[r2, c2], d8[r2 + 1, c2 + 1], d8[r2 + 1, c2] = 2, 16, 64  # SE -> W -> N
    return d8


def cases():
    out = {}
    out["flwdir0"] = dict(d8=np.loadtxt(os.path.join(REF, "tests/data/flwdir.asc"), dtype=np.uint8),
                          src="reference tests/data/flwdir.asc (tests/conftest.py:19-20)")
    out["flwdir_large"] = dict(d8=np.loadtxt(os.path.join(REF, "tests/data/flwdir1.asc"), dtype=np.uint8),
                               src="reference tests/data/flwdir1.asc (tests/conftest.py:116-118)")
    np.random.seed(2345)
    out["flwdir1"] = dict(d8=pyflwdir.from_dem(np.random.rand(15, 10)).to_array("d8"),
                          src="from_dem(np.random.rand(15,10)), seed 2345 (tests/conftest.py:57-60)")
    d8, tr = read_tif(os.path.join(REF, "examples/rhine_d8.tif"))
    elv, _ = read_tif(os.path.join(REF, "examples/rhine_elv0.tif"))
    out["rhine"] = dict(d8=d8.astype(np.uint8), elevtn=elv.astype(np.float32), transform=tr, latlon=True,
                        src="reference examples/rhine_d8.tif + rhine_elv0.tif")
    for name, shape, seed, kw in [
        ("synth_river_256", (256, 256), 0, dict(O.SYNTH_RIVER)),
        ("synth_rough_nodata_384x512", (384, 512), 1, dict(O.SYNTH_ROUGH, nodata_pct=30)),
        ("synth_river_nodata_768x1024", (768, 1024), 2, dict(O.SYNTH_RIVER, nodata_pct=25)),
        ("synth_loops_96x80", (96, 80), 3, dict(O.SYNTH_ROUGH, nodata_pct=20)),
        ("synth_tiny_5x7", (5, 7), 4, dict(O.SYNTH_ROUGH)),
        ("synth_onerow_1x300", (1, 300), 5, dict(O.SYNTH_ROUGH)),
        ("synth_onecol_300x1", (300, 1), 6, dict(O.SYNTH_ROUGH)),
    ]:
        d8 = O.synth_d8(shape[0], shape[1], seed=seed, **kw)
        if "loops" in name:
            d8 = inject_loops(d8)
        out[name] = dict(d8=d8, elevtn=O.synth_elev_f32(shape[0], shape[1], seed=seed, **kw),
                         synth=dict(seed=seed, nrow=shape[0], ncol=shape[1], **kw),
                         src="oracle.synth_d8" + (" + injected loops" if "loops" in name else ""))
    return out


def run_case(name, case):
    d8 = np.ascontiguousarray(case["d8"], dtype=np.uint8)
    nrow, ncol = d8.shape
    n = d8.size
    tr = case.get("transform", GI.DEFAULT_TRANSFORM)
    latlon = case.get("latlon", True)
    res = {}
    t0 = time.time()

    # ---- kernel level (L1/L0 free functions) -----------------------------------------
    for dt in (np.int32, np.uint32, np.int64):
        if dt is not np.int32 and n > FULL_LIMIT:
            continue
        sfx = np.dtype(dt).name
        idxs_ds, idxs_pit, nvalid = core_d8.from_array(d8, dtype=dt)
        mv = dt(core._mv) if dt is not np.int64 else core._mv
        seq = core.idxs_seq(idxs_ds, idxs_pit, mv)
        res[f"idxs_ds_{sfx}"] = idxs_ds
        res[f"idxs_pit_{sfx}"] = idxs_pit
        res[f"idxs_seq_{sfx}"] = seq
        if dt is np.int32:
            res["n_valid"] = np.int64(nvalid)
            res["n_upstream"] = core.upstream_count(idxs_ds, mv)
            rnk, nrank = core.rank(idxs_ds, mv)
            res["rank"] = rnk
            res["n_rank"] = np.int64(nrank)

    # ---- API level ------------------------------------------------------------------
    A = Affine(*tr)
    flw = pyflwdir.from_array(d8, ftype="d8", check_ftype=False, transform=A, latlon=latlon, cache=False)
    flw_proj = pyflwdir.from_array(d8, ftype="d8", check_ftype=False,
                                   transform=Affine(*GI.PROJ_TRANSFORM), latlon=False, cache=False)
    res["idxs_outlet"] = flw.idxs_outlet
    upa = flw.upstream_area()
    res["uparea_cell"] = upa
    res["uparea_km2_latlon"] = flw.upstream_area("km2")
    res["uparea_ha_proj"] = flw_proj.upstream_area("ha")
    P = GI.payloads(d8.shape)
    res["accuflux_f32"] = flw.accuflux(P["w32"])
    res["accuflux_f64"] = flw.accuflux(P["w64"])
    res["accuflux_ds_f32"] = flw.accuflux(P["w32"], direction="down")
    res["accuflux_i32_nodata"] = flw.accuflux(P["wi32_nodata"], nodata=-9999)
    res["accuflux_ds_i32_nodata"] = flw.accuflux(P["wi32_nodata"], nodata=-9999, direction="down")
    res["accuflux_f32_nodata_m1"] = flw.accuflux(P["wf32_nodata_m1"], nodata=-1)
    res["accuflux_i64"] = flw.accuflux(P["wi64"])
    res["strahler"] = flw.stream_order()
    thr = GI.threshold(upa)
    res["strahler_mask_upa"] = flw.stream_order(mask=upa > thr)
    res["strahler_mask_rand"] = flw.stream_order(mask=GI.random_mask(d8.shape))
    res["basins"] = flw.basins()
    idxs, ids = GI.basin_outlets(upa, flw.idxs_pit)
    res["basins_idxs"] = idxs
    res["basins_ids"] = ids
    res["basins_sub_i16"] = flw.basins(idxs=idxs, ids=ids)
    elevtn = GI.elevation(case.get("elevtn"), upa)
    drain = upa > thr
    res["hand_f32"] = flw.hand(drain, elevtn)
    res["hand_f64"] = flw.hand(drain, elevtn.astype(np.float64) * 1.000001)
    res["hand_thr"] = np.int64(thr)
    # ---- main upstream cell, classic stream order, stream distance ----------
    res["idxs_us_main"] = flw.idxs_us_main                                   # uparea = upstream_area() int32
    res["idxs_us_main_km2"] = flw.main_upstream(uparea=res["uparea_km2_latlon"])  # float64 areas
    res["strord_classic"] = flw.stream_order(type="classic")
    res["strord_classic_mask"] = flw.stream_order(type="classic", mask=upa > thr)
    res["strdist_cell"] = flw.stream_distance(unit="cell")
    res["strdist_cell_mask"] = flw.stream_distance(mask=upa > thr, unit="cell")
    res["strdist_m_latlon"] = flw.stream_distance(unit="m")
    res["strdist_m_proj"] = flw_proj.stream_distance(unit="m")
    res["strdist_m_mask"] = flw.stream_distance(mask=GI.random_mask(d8.shape), unit="m")
    # ---- codecs — re-encoding to D8 / LDD and an LDD raster as input -----------------
    res["to_array_d8"] = flw.to_array("d8")
    ldd = flw.to_array("ldd")
    res["to_array_ldd"] = ldd
    try:
        flw_ldd = pyflwdir.from_array(ldd, ftype="infer", transform=A, latlon=latlon, cache=False)
        res["ldd_inferred_is_ldd"] = np.uint8(flw_ldd.ftype == "ldd")
        res["ldd_idxs_ds"] = flw_ldd.idxs_ds
        res["ldd_idxs_outlet"] = flw_ldd.idxs_outlet
        res["ldd_uparea_cell"] = flw_ldd.upstream_area()
    except OverflowError:
        # interpreted (no JIT) under numpy >= 2 the reference's own core_ldd.from_array overflows its
        # int8 column offset on rasters wider than 127 columns (core_ldd.py:55): LDD *input* is pinned
        # on the narrow cases only; the LDD *output* (to_array) on all of them
        pass

    stats = dict(shape=[int(nrow), int(ncol)], n_valid=int(res["n_valid"]), n_pits=int(res["idxs_pit_int32"].size),
                 n_seq=int(res["idxs_seq_int32"].size), max_rank=int(res["rank"].max()),
                 n_loop_cells=int((res["rank"] == -1).sum()), uparea_max=int(upa.max()),
                 strahler_max=int(res["strahler"].max()),
                 indegree_hist=np.bincount(res["n_upstream"][res["n_upstream"] >= 0], minlength=9).tolist(),
                 ref_seconds=round(time.time() - t0, 2))
    return res, stats, elevtn, tr, latlon


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    mpath = os.path.join(GOLD, "manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    manifest["_meta"] = dict(reference="Deltares/pyflwdir " + pyflwdir.__version__,
                             mode="interpreted (identity-njit shim == NUMBA_DISABLE_JIT=1)",
                             numpy=np.__version__, generator="oracle/gen_golden.py")
    for name, case in cases().items():
        if args.only and name != args.only:
            continue
        print(f"[gen_golden] {name} {case['d8'].shape} ...", flush=True)
        res, stats, elevtn, tr, latlon = run_case(name, case)
        entry = dict(src=case["src"], stats=stats, transform=list(tr), latlon=bool(latlon),
                     synth=case.get("synth"), digests={k: digest(np.asarray(v)) for k, v in res.items()},
                     dtypes={k: np.asarray(v).dtype.str for k, v in res.items()})
        store = dict(d8=case["d8"])
        if "elevtn" in case and "synth" not in case:
            store["elevtn"] = case["elevtn"]  # real-world elevation cannot be regenerated
        if case["d8"].size <= FULL_LIMIT:
            store.update({f"out_{k}": np.asarray(v) for k, v in res.items()})
            entry["full"] = True
        else:  # inputs that are needed to re-run but cheap to keep
            store.update({f"out_{k}": np.asarray(res[k]) for k in ("basins_idxs", "basins_ids", "hand_thr")})
            entry["full"] = False
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **store)
        manifest[name] = entry
        print(f"    {stats}", flush=True)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
