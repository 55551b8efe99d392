#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — golden vectors of the rows SURVEY.md 8(f) marks "next" (the callers and data formats
either side of the hot path), recorded from the imported reference exactly like oracle/gen_golden.py does for
the path itself (interpreted through the identity-njit shim; runs only where /root/reference exists).

    tests/golden/wide_arith.npz    Flwdir.upstream_sum                        (reference pyflwdir/arithmetics.py:147-169)
    tests/golden/wide_dem.npz      dem.fill_depressions / from_dem            (reference pyflwdir/dem.py:17-143)
    tests/golden/wide_general.npz  NEXTXY rasters and FlwdirRaster(idxs_ds=...) with arbitrary links, all operations
    tests/golden/wide_snap.npz     Flwdir.snap, basins / add_pits with streams=
    tests/golden/wide_subgrid.npz  FlwdirRaster.ucat_area / .floodplains      (subgrid.py:51-93, dem.py:333-379)

Fixtures are data only (inputs + the reference's outputs).   Usage: python oracle/gen_golden_wide.py [dem ...]
"""
from __future__ import annotations

import json
import os
import sys

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
from pyflwdir import dem as rdem  # noqa: E402

from oracle import oracle as O  # noqa: E402  (synthetic inputs only)

GOLD = os.path.join(ROOT, "tests", "golden")

WANG_LIU = [[15, 15, 14, 15, 12, 6, 12], [14, 13, 10, 12, 15, 17, 15], [15, 15, 9, 11, 8, 15, 15],
            [16, 17, 8, 16, 15, 7, 5], [19, 18, 19, 18, 17, 15, 14]]  # reference tests/test_dem.py:13-21


def gen_dem():
    store = {}
    calls = []  # (key, input key, kwargs)

    def run(key, inp_key, **kw):
        a = store["in_" + inp_key]
        try:
            filled, d8 = rdem.fill_depressions(a.copy(), **kw)
            store[f"out_{key}_elev"] = filled
            store[f"out_{key}_d8"] = d8
            calls.append((key, inp_key, kw, "ok"))
        except (IndexError, ValueError) as exc:
            calls.append((key, inp_key, kw, type(exc).__name__))

    for dt in (np.float32, np.int32, np.float64):
        nm = "wl_" + np.dtype(dt).name
        store["in_" + nm] = np.array(WANG_LIU, dtype=dt)
        run(nm + "_default", nm)
        run(nm + "_min", nm, outlets="min")
        run(nm + "_maxdepth2", nm, max_depth=2)
        run(nm + "_conn4", nm, connectivity=4)
        run(nm + "_pits", nm, idxs_pit=np.array([5, 27]))
        run(nm + "_elvmax", nm, elv_max=13)
    a = np.array(WANG_LIU, dtype=np.float32)
    a[3, 5:] = -9999
    store["in_wl_nodata"] = a
    run("wl_nodata", "wl_nodata")
    a = np.array(WANG_LIU, dtype=np.float32)
    a[0, 0] = np.nan
    a[2, 3] = np.nan
    store["in_wl_nan"] = a
    run("wl_nan", "wl_nan", nodata=np.nan)
    np.random.seed(2345)
    store["in_rand_15x10"] = np.random.rand(15, 10)  # float64: reference tests/conftest.py:57-60
    run("rand_15x10", "rand_15x10")
    run("rand_15x10_min", "rand_15x10", outlets="min")
    rng = np.random.default_rng(7)
    store["in_rand_40x50_f32"] = (rng.random((40, 50)) * 100).astype(np.float32)
    run("rand_40x50", "rand_40x50_f32")
    run("rand_40x50_md5", "rand_40x50_f32", max_depth=5.0)
    run("rand_40x50_md20", "rand_40x50_f32", max_depth=20.0)
    run("rand_40x50_conn4", "rand_40x50_f32", connectivity=4)
    flat = np.round(rng.random((30, 36)) * 4).astype(np.float32)  # many ties: the heap's tie-breaking decides the D8
    store["in_flat_30x36"] = flat
    run("flat_30x36", "flat_30x36")
    e = O.synth_elev_f32(120, 160, seed=3, tilt=100000, white=2, nodata_pct=20)
    d8s = O.synth_d8(120, 160, seed=3, tilt=100000, white=2, nodata_pct=20)
    e = np.where(d8s == 247, np.float32(-9999), e).astype(np.float32)
    store["in_synth_120x160"] = e
    run("synth_120x160", "synth_120x160")
    def enc(v):  # JSON, so that the test never has to eval() anything
        if isinstance(v, np.ndarray):
            return {"__array__": v.tolist()}
        if isinstance(v, float) and np.isnan(v):
            return {"__nan__": True}
        return v

    store["calls"] = np.array(json.dumps([[k, i, {a: enc(b) for a, b in kw.items()}, st] for k, i, kw, st in calls]))
    np.savez_compressed(os.path.join(GOLD, "wide_dem.npz"), **store)
    print(f"[wide] dem: {len(calls)} calls:", [(c[0], c[3]) for c in calls if c[3] != 'ok'] or "all ok")


def gen_subgrid():
    """FlwdirRaster.ucat_area (pyflwdir.py:1159-1191, subgrid.py:51-93) and .floodplains (pyflwdir.py:1513-1545,
    dem.py:333-379) on the golden rasters of gen_golden.py (their D8 / elevation inputs are read back from
    tests/golden/<case>.npz)."""
    from pyflwdir_amd._affine import Affine
    from oracle import golden_inputs as GI
    import json

    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    store = {}
    for name, cellsize in [("flwdir0", 5), ("flwdir_large", 10), ("flwdir1", 3), ("synth_loops_96x80", 8), ("rhine", 20),
                           ("synth_rough_nodata_384x512", 16)]:
        z = np.load(os.path.join(GOLD, name + ".npz"))
        d8 = z["d8"]
        ent = manifest[name]
        for tag, tr, latlon in (("ll", ent["transform"], ent["latlon"]), ("pr", GI.PROJ_TRANSFORM, False)):
            flw = pyflwdir.from_array(d8, ftype="d8", check_ftype=False, transform=Affine(*tr), latlon=latlon, cache=False)
            upa = flw.upstream_area()
            if tag == "ll":
                idxs_out = flw.ucat_outlets(cellsize, uparea=upa)[0]
                # + a repeated outlet and a missing one: the reference's loop semantics for both
                io = idxs_out.copy()
                flat = io.ravel()
                valid = np.flatnonzero(flat != flw._mv)
                if valid.size > 3:
                    flat[valid[1]] = flat[valid[0]]
                store[f"in_{name}_idxs_out"] = io
                store[f"in_{name}_idxs_out_dup"] = flat.reshape(io.shape)
            for key_io in ("idxs_out", "idxs_out_dup"):
                io = store[f"in_{name}_{key_io}"]
                for unit in ("cell", "km2", "ha"):
                    m, a = flw.ucat_area(io, unit=unit)
                    store[f"out_{name}_{tag}_{key_io}_{unit}_map"] = m
                    store[f"out_{name}_{tag}_{key_io}_{unit}_are"] = a
            if "elevtn" in z.files:
                elv = z["elevtn"].astype(np.float32)
            elif ent.get("synth"):
                sy = ent["synth"]
                elv = O.synth_elev_f32(sy["nrow"], sy["ncol"], seed=sy["seed"], tilt=sy["tilt"], white=sy["white"],
                                       nodata_pct=sy["nodata_pct"])
            else:
                elv = GI.elevation(None, upa)
            store[f"in_{name}_elevtn"] = elv
            upa_km2 = flw.upstream_area("km2")
            thr = float(np.percentile(upa_km2[upa_km2 > 0], 90)) if (upa_km2 > 0).any() else 1.0
            store[f"in_{name}_{tag}_upa_min"] = np.float64(thr)
            store[f"out_{name}_{tag}_fld"] = flw.floodplains(elv, upa_min=thr, b=0.3)
            store[f"out_{name}_{tag}_fld_b05_f64"] = flw.floodplains(elv.astype(np.float64) * 1.000001, uparea=upa_km2,
                                                                      upa_min=thr * 0.5, b=0.5)
    np.savez_compressed(os.path.join(GOLD, "wide_subgrid.npz"), **store)
    print(f"[wide] subgrid: {len(store)} arrays")


def gen_snap():
    """Flwdir.snap (downstream, cell units) and basins(xy/idxs, streams=...) / add_pits(streams=...):
    reference pyflwdir/flwdir.py:404-463,805-811, core.py:440-480."""
    from pyflwdir_amd._affine import Affine
    import json

    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    store = {}
    # (rasters without cycles only: the reference's trace never returns from a cycle)
    for name in ("flwdir0", "flwdir_large", "synth_rough_nodata_384x512", "rhine"):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        d8 = z["d8"]
        ent = manifest[name]
        flw = pyflwdir.from_array(d8, ftype="d8", check_ftype=False, transform=Affine(*ent["transform"]),
                                  latlon=ent["latlon"], cache=False)
        upa = flw.upstream_area()
        strord = flw.stream_order()
        streams = strord >= max(2, int(strord.max()) - 2)
        rng = np.random.default_rng(11)
        valid = np.flatnonzero(flw.mask)
        idxs = rng.choice(valid, size=min(200, valid.size), replace=False).astype(np.int64)
        store[f"in_{name}_idxs"] = idxs
        store[f"in_{name}_streams"] = streams
        i1, d1 = flw.snap(idxs=idxs, mask=streams)
        store[f"out_{name}_snap_idxs"], store[f"out_{name}_snap_dist"] = i1, d1
        i2, d2 = flw.snap(idxs=idxs, mask=streams, max_length=5)
        store[f"out_{name}_snap5_idxs"], store[f"out_{name}_snap5_dist"] = i2, d2
        i3, d3 = flw.snap(idxs=idxs)
        store[f"out_{name}_snap_nomask_idxs"], store[f"out_{name}_snap_nomask_dist"] = i3, d3
        store[f"out_{name}_basins_streams"] = flw.basins(idxs=idxs[:40], streams=streams)
        flw2 = pyflwdir.from_array(d8, ftype="d8", check_ftype=False, cache=False)
        flw2.add_pits(idxs=idxs[:10], streams=streams)
        store[f"out_{name}_addpits_streams_idxs_pit"] = flw2.idxs_pit
        store[f"out_{name}_addpits_streams_upa"] = flw2.upstream_area()
    np.savez_compressed(os.path.join(GOLD, "wide_snap.npz"), **store)
    print(f"[wide] snap: {len(store)} arrays")


def gen_general():
    """Graphs with links outside the 8 neighbours: NEXTXY rasters (core_nextxy.py:24-86, from_array / to_array;
    pyflwdir.py:170-205) and FlwdirRaster(idxs_ds=...) with arbitrary links (pyflwdir.py:211-273, as upscale
    builds them, :1079-1085) — every operation of the path on both; plus the keys of the dump/load dictionary
    (pyflwdir.py:275-286)."""
    from pyflwdir_amd._affine import Affine
    from oracle import golden_inputs as GI
    import json

    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    store = {}

    def ops(tag, flw, elv):
        upa = flw.upstream_area()
        store[f"out_{tag}_idxs_ds"] = flw.idxs_ds
        store[f"out_{tag}_idxs_pit"] = flw.idxs_pit
        if flw.idxs_outlet is not None:  # (None for FlwdirRaster(idxs_ds=...) built without outlets)
            store[f"out_{tag}_idxs_outlet"] = flw.idxs_outlet
        store[f"out_{tag}_idxs_seq"] = flw.idxs_seq
        store[f"out_{tag}_rank"] = flw.rank
        store[f"out_{tag}_n_upstream"] = flw.n_upstream
        store[f"out_{tag}_upa"] = upa
        store[f"out_{tag}_upa_km2"] = flw.upstream_area("km2")
        P = GI.payloads(flw.shape)
        store[f"out_{tag}_accu_f32"] = flw.accuflux(P["w32"])
        store[f"out_{tag}_accu_ds_f64"] = flw.accuflux(P["w64"], direction="down")
        store[f"out_{tag}_accu_i32_nd"] = flw.accuflux(P["wi32_nodata"], nodata=-9999)
        store[f"out_{tag}_strahler"] = flw.stream_order()
        store[f"out_{tag}_strahler_mask"] = flw.stream_order(mask=GI.random_mask(flw.shape))
        store[f"out_{tag}_classic"] = flw.stream_order(type="classic")
        store[f"out_{tag}_idxs_us_main"] = flw.idxs_us_main
        store[f"out_{tag}_basins"] = flw.basins()
        idxs, ids = GI.basin_outlets(upa, flw.idxs_pit)
        store[f"in_{tag}_basins_idxs"], store[f"in_{tag}_basins_ids"] = idxs, ids
        store[f"out_{tag}_basins_sub"] = flw.basins(idxs=idxs, ids=ids)
        thr = GI.threshold(upa)
        store[f"out_{tag}_hand"] = flw.hand(upa > thr, elv)
        store[f"out_{tag}_dist_cell"] = flw.stream_distance(unit="cell")
        store[f"out_{tag}_dist_m"] = flw.stream_distance(unit="m")
        store[f"out_{tag}_dist_m_mask"] = flw.stream_distance(mask=upa > thr, unit="m")
        flw.order_cells(method="sort")
        store[f"out_{tag}_idxs_seq_sort"] = flw.idxs_seq

    for name in ("flwdir0", "flwdir_large", "synth_loops_96x80"):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        ent = manifest[name]
        A = Affine(*ent["transform"])
        flw8 = pyflwdir.from_array(z["d8"], ftype="d8", check_ftype=False, transform=A, latlon=ent["latlon"], cache=False)
        upa8 = flw8.upstream_area()
        elv = GI.elevation(None, upa8)
        store[f"in_{name}_elevtn"] = elv
        # NEXTXY: the same graph written as a NEXTXY raster and parsed back
        nxy = flw8.to_array("nextxy")
        store[f"in_{name}_nextxy"] = nxy
        flwn = pyflwdir.from_array(nxy, ftype="nextxy", transform=A, latlon=ent["latlon"], cache=False)
        store[f"out_{name}_nextxy_isvalid"] = np.uint8(pyflwdir.core_nextxy.isvalid(nxy))
        ops(name + "_nextxy", flwn, elv)
        store[f"out_{name}_nextxy_to_array"] = flwn.to_array()
        store[f"out_{name}_nextxy_to_d8"] = flwn.to_array("d8")
        # arbitrary links: every cell skips its downstream cell (the forest of 2-step links)
        ds = flw8.idxs_ds
        mv = flw8._mv
        ds2 = np.where(ds == mv, mv, ds[np.where(ds == mv, 0, ds)]).astype(ds.dtype)
        store[f"in_{name}_ds2"] = ds2
        flw2 = pyflwdir.FlwdirRaster(idxs_ds=ds2, shape=flw8.shape, ftype="d8", transform=A, latlon=ent["latlon"], cache=False)
        ops(name + "_ds2", flw2, elv)
        store[f"out_{name}_ds2_to_nextxy"] = flw2.to_array("nextxy")
    # from_array(check_ftype=False) on values outside the D8 alphabet (decoded by core_d8.drdc, core_d8.py:20-37)
    rng = np.random.default_rng(3)
    for nm, vals in (("odd_nbr", [3, 5, 6, 7, 17, 33, 65, 127, 129, 200, 4, 4, 4, 2, 8, 1, 16, 64, 247, 0]),
                     ("odd_far", [9, 12, 15, 4, 4, 2, 8, 1, 16, 64, 32, 128, 247, 0, 255])):
        d = rng.choice(np.array(vals, np.uint8), size=(40, 50))
        d[-1, :] = 0
        store[f"in_{nm}"] = d
        f = pyflwdir.from_array(d, ftype="d8", check_ftype=False, cache=False)
        store[f"out_{nm}_idxs_ds"] = f.idxs_ds
        store[f"out_{nm}_idxs_pit"] = f.idxs_pit
        store[f"out_{nm}_idxs_outlet"] = f.idxs_outlet
        store[f"out_{nm}_rank"] = f.rank
    store["dump_keys"] = np.array(",".join(sorted(str(k) for k in flw2._dict.keys())))
    np.savez_compressed(os.path.join(GOLD, "wide_general.npz"), **store)
    print(f"[wide] general: {len(store)} arrays")


def gen_general_pits():
    """add_pits on general graphs (reference pyflwdir/flwdir.py:261-279 resets the cell order; the next sweep
    re-orders — NEXTXY rasters by rank, pyflwdir.py:292-297): a NEXTXY raster and a raster of non-neighbour links,
    three interior cells turned into pits, then the order and the order-sensitive sweeps."""
    from pyflwdir_amd._affine import Affine
    from oracle import golden_inputs as GI
    import json

    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    W = np.load(os.path.join(GOLD, "wide_general.npz"))
    store = {}
    for name in ("flwdir0", "flwdir_large", "synth_loops_96x80"):
        ent = manifest[name]
        A = Affine(*ent["transform"])
        elv = W[f"in_{name}_elevtn"]
        nxy = W[f"in_{name}_nextxy"]
        ds2 = W[f"in_{name}_ds2"]
        for kind in ("nextxy", "ds2"):
            if kind == "nextxy":
                flw = pyflwdir.from_array(nxy, ftype="nextxy", transform=A, latlon=ent["latlon"], cache=False)
            else:
                flw = pyflwdir.FlwdirRaster(idxs_ds=ds2, shape=nxy.shape[1:], ftype="d8", transform=A, latlon=ent["latlon"], cache=False)
            tag = f"{name}_{kind}"
            upa0 = flw.upstream_area()
            flw.order_cells(method="sort")  # (an installed order exists when the pits arrive)
            cand = W[f"in_{tag}_basins_idxs"]  # a few pits + high-accumulation interior cells
            idxs = cand[~np.isin(cand, flw.idxs_pit)][:3]  # three interior cells with a large upstream area
            assert idxs.size == 3
            store[f"in_{tag}_pits"] = idxs
            flw.add_pits(idxs=idxs)
            store[f"out_{tag}_idxs_pit"] = flw.idxs_pit
            store[f"out_{tag}_idxs_ds"] = flw.idxs_ds
            store[f"out_{tag}_idxs_seq"] = flw.idxs_seq
            store[f"out_{tag}_rank"] = flw.rank
            upa = flw.upstream_area()
            assert not np.array_equal(upa, upa0)
            store[f"out_{tag}_upa"] = upa
            P = GI.payloads(flw.shape)
            store[f"out_{tag}_accu_f32"] = flw.accuflux(P["w32"])
            store[f"out_{tag}_accu_ds_f64"] = flw.accuflux(P["w64"], direction="down")
            store[f"out_{tag}_strahler"] = flw.stream_order()
            store[f"out_{tag}_basins"] = flw.basins()
            store[f"out_{tag}_hand"] = flw.hand(upa > GI.threshold(upa), elv)
    np.savez_compressed(os.path.join(GOLD, "wide_general_pits.npz"), **store)
    print(f"[wide] general_pits: {len(store)} arrays")


def gen_snap2():
    """Flwdir.snap upstream (along the main upstream cells) and in metres, with and without mask / max_length
    (reference pyflwdir/flwdir.py:500-560, core.snap / core._trace core.py:308-366,440-480); accuflux of narrow
    integer payloads (int8 / int16 / uint8 / uint16: the reference accumulates in the payload's own dtype,
    streams.py:36 `data.copy()`)."""
    from pyflwdir_amd._affine import Affine
    import json

    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    W = np.load(os.path.join(GOLD, "wide_snap.npz"))
    store = {}
    for name in ("flwdir0", "flwdir_large", "synth_rough_nodata_384x512", "rhine"):
        z = np.load(os.path.join(GOLD, name + ".npz"))
        ent = manifest[name]
        flw = pyflwdir.from_array(z["d8"], ftype="d8", check_ftype=False, transform=Affine(*ent["transform"]),
                                  latlon=ent["latlon"], cache=False)
        idxs, streams = W[f"in_{name}_idxs"], W[f"in_{name}_streams"]
        upa = flw.upstream_area()
        heads = upa <= 2  # a mask for the upstream direction: stop at (near-)headwater cells
        store[f"in_{name}_heads"] = heads
        cases = dict(down_m=dict(mask=streams, unit="m"), down_m_max=dict(mask=streams, unit="m", max_length=2500.0),
                     down_m_nomask=dict(unit="m"), up_cell=dict(direction="up"), up_cell_mask=dict(direction="up", mask=heads),
                     up_cell_max=dict(direction="up", max_length=7), up_m=dict(direction="up", unit="m"),
                     up_m_max=dict(direction="up", unit="m", mask=heads, max_length=4000.0))
        for key, kw in cases.items():
            i1, d1 = flw.snap(idxs=idxs, **kw)
            store[f"out_{name}_{key}_idxs"], store[f"out_{name}_{key}_dist"] = i1, d1
        if name in ("flwdir0", "flwdir_large"):
            rng = np.random.default_rng(23)
            for dt, lo, hi in ((np.int8, -3, 4), (np.int16, -20, 60), (np.uint8, 0, 3), (np.uint16, 0, 40)):
                data = rng.integers(lo, hi, size=z["d8"].shape).astype(dt)
                nm = np.dtype(dt).name
                store[f"in_{name}_{nm}"] = data
                with np.errstate(over="ignore"):
                    store[f"out_{name}_{nm}_up"] = flw.accuflux(data, nodata=2)
                    store[f"out_{name}_{nm}_down"] = flw.accuflux(data, nodata=-9999, direction="down")
    np.savez_compressed(os.path.join(GOLD, "wide_snap2.npz"), **store)
    print(f"[wide] snap2: {len(store)} arrays")


GENS = {"dem": gen_dem, "subgrid": gen_subgrid, "snap": gen_snap, "snap2": gen_snap2, "general": gen_general,
        "general_pits": gen_general_pits}

def gen_arith():
    """Flwdir.upstream_sum (reference pyflwdir/flwdir.py:412-433, arithmetics.py:147-169) for int32 / int64 /
    float32 / float64 data with missing values in it, on the golden rasters."""
    from pyflwdir_amd._affine import Affine  # noqa: F401
    store = {}
    rng = np.random.default_rng(11)
    for name in ("flwdir0", "flwdir1", "synth_loops_96x80", "synth_tiny_5x7", "synth_onerow_1x300"):
        fn = os.path.join(GOLD, name + ".npz")
        if not os.path.exists(fn):
            continue
        d8 = np.load(fn)["d8"]
        flw = pyflwdir.from_array(d8, ftype="d8", check_ftype=False)
        for dt in ((np.int32, np.float32) if name == "flwdir1" else (np.int32, np.int64, np.float32, np.float64)):
            if np.dtype(dt).kind == "i":
                data = rng.integers(-50, 1000, size=d8.shape).astype(dt)
            else:
                data = (rng.random(d8.shape) * 100).astype(dt)
            data[rng.random(d8.shape) < 0.05] = -9999  # missing values inside the domain
            key = f"{name}_{np.dtype(dt).name}"
            store["in_" + key] = data
            store["out_" + key] = flw.upstream_sum(data, mv=-9999)
        data = np.ones(d8.shape, np.float64)
        store[f"out_{name}_ones_nan"] = flw.upstream_sum(data, mv=np.nan)  # a missing value that never compares equal
    store["cases"] = np.array(repr(sorted(k[3:] for k in store if k.startswith("in_"))))
    np.savez_compressed(os.path.join(GOLD, "wide_arith.npz"), **store)
    print(f"[wide] arith: {len(store)} arrays")


GENS["arith"] = gen_arith

if __name__ == "__main__":
    for name in (sys.argv[1:] or GENS):
        GENS[name]()
