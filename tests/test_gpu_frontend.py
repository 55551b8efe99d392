"""The path a drop-in numpy caller takes: the front end's default arena, the device-side alphabet
check of from_array on large rasters, and the library's own account of its host <-> device traffic
(pfd_transfer_stats; reference pyflwdir/pyflwdir.py:130-205, 770-801).  The allocator is process-wide state: child processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip
mode = sys.argv[1]
res = {}
if mode == "explicit":
    _hip.reserve(256 << 20)
d8 = O.synth_d8(4099, 4231, seed=5, tilt=1 << 26, white=2, nodata_pct=3)
exp = O.upstream_area_cell(d8)[0]
free0 = _hip.mem_info()["free"]
s0 = _hip.alloc_stats()
t0 = time.perf_counter()
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
upa = flw.upstream_area()
t1 = time.perf_counter()
s1 = _hip.alloc_stats()
so = flw.stream_order()
acc = flw.accuflux(np.ones(d8.shape, np.float32))
s2 = _hip.alloc_stats()
del flw
t2 = time.perf_counter()
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
upa2 = flw.upstream_area()
t3 = time.perf_counter()
res.update(ok=bool(np.array_equal(upa, exp) and np.array_equal(upa2, exp)), n=int(d8.size), free0=free0,
           reserved=s1["reserved_bytes"], arena_blocks=s2["arena_blocks"] - s0["arena_blocks"],
           hipmalloc=s2["hipmalloc_calls"] - s0["hipmalloc_calls"], first_ms=(t1 - t0) * 1e3, second_ms=(t3 - t2) * 1e3,
           acc_ok=bool(np.array_equal(acc[d8 != 247].astype(np.int64), upa[d8 != 247])))
del flw
_hip.reserve(0)
res["reserved_after_release"] = _hip.alloc_stats()["reserved_bytes"]
print(json.dumps(res))
''' % ROOT


def _child(mode, env=None):
    e = dict(os.environ)
    e.pop("PFD_RESERVE_GIB", None)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_front_end_reserves_a_default_arena(gpu_lib):
    """No explicit pfd_reserve: the first handle of the process reserves min(0.5 x free HBM, 48 B x n_cells) and the
    working buffers (>= 1 MiB) of every operation come out of it — the allocator bench.py times is the default one."""
    r = _child("default")
    assert r["ok"] and r["acc_ok"]
    want = min(r["free0"] // 2, 48 * r["n"])
    assert abs(r["reserved"] - want) <= (2 << 20), (r["reserved"], want)
    assert r["arena_blocks"] > 10  # the tile records, the plan, the staging buffers ...
    assert r["hipmalloc"] < 64, r  # (blocks below 1 MiB stay with hipMalloc + the class cache: a few dozen small calls)
    assert r["reserved_after_release"] == 0
    print(f"first from_array + upstream_area {r['first_ms']:.1f} ms, on a second handle {r['second_ms']:.1f} ms")


def test_default_arena_stands_down(gpu_lib):
    """PFD_RESERVE_GIB=0 switches the default arena off; a process that reserved explicitly keeps its own size."""
    r = _child("default", {"PFD_RESERVE_GIB": "0"})
    assert r["ok"] and r["reserved"] == 0 and r["arena_blocks"] == 0
    r = _child("default", {"PFD_RESERVE_GIB": "0.5"})
    assert r["ok"] and r["reserved"] == 512 << 20
    r = _child("explicit")
    assert r["ok"] and r["reserved"] == 256 << 20


def test_from_array_checks_large_rasters_on_the_device(gpu_lib, oracle):
    """From 2**24 cells on, "is every byte a D8 value" (reference core_d8.isvalid, pyflwdir.py:181-182) is answered by the
    device pass that builds the graph: same errors, same inference."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import raster

    d8 = oracle.synth_d8(4100, 4100, seed=1, tilt=100000, white=2, nodata_pct=5)
    assert d8.size >= raster._INFER_ON_DEVICE_MIN
    exp = oracle.upstream_area_cell(d8)[0]
    for ftype in ("d8", "infer"):
        flw = pyflwdir.from_array(d8, ftype=ftype)
        assert flw.ftype == "d8" and np.array_equal(flw.upstream_area(), exp)
    bad = d8.copy()
    bad[2000, 17] = 3
    with pytest.raises(ValueError, match='The flow direction data with type "d8" is invalid.'):
        pyflwdir.from_array(bad, ftype="d8")
    with pytest.raises(ValueError, match="could not be inferred"):
        pyflwdir.from_array(bad, ftype="infer")
    # an LDD raster of that size is still inferred as LDD (the device says "not D8", the host alphabets decide)
    ldd = raster._D8_TO_LDD[d8]
    flw = pyflwdir.from_array(ldd, ftype="infer")
    assert flw.ftype == "ldd" and np.array_equal(flw.upstream_area(), exp)
    nopit = np.full((4100, 4100), 1, np.uint8)
    nopit[:, -1] = 16  # every cell flows east, the last column back west: a raster of two-cell loops without a pit
    with pytest.raises(ValueError, match="no pits found"):
        pyflwdir.from_array(nopit, ftype="d8")


def test_transfer_stats_tell_upload_and_download_apart(gpu_lib, oracle):
    """pfd_transfer_stats: bytes and milliseconds of the staging copies of this thread's calls (H2D / D2H
    reported separately); a large host result is pre-faulted while the kernels run and arrives unchanged."""
    import pyflwdir_amd as pyflwdir
    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(5000, 5000, seed=2, tilt=1 << 26, white=2, nodata_pct=0)
    _hip.transfer_stats(reset=True)
    flw = pyflwdir.from_array(d8, ftype="d8")
    a = _hip.transfer_stats(reset=True)
    assert a["h2d_bytes"] >= d8.size and a["h2d_ms"] > 0
    upa = flw.upstream_area()  # 100 MB result: above the pre-fault threshold
    b = _hip.transfer_stats(reset=True)
    assert b["d2h_bytes"] == upa.nbytes and b["d2h_ms"] > 0 and b["host_results"] == 1 and b["h2d_bytes"] == 0
    assert np.array_equal(upa, oracle.upstream_area_cell(d8)[0])
    w = np.random.default_rng(3).random(d8.shape).astype(np.float32)
    acc = flw.accuflux(w)
    c = _hip.transfer_stats(reset=True)
    assert c["h2d_bytes"] == w.nbytes and c["d2h_bytes"] == acc.nbytes
    idxs_ds, idxs_pit, _ = oracle.from_array(d8)
    assert np.array_equal(acc.ravel(), oracle.accuflux(idxs_ds, oracle.idxs_seq(idxs_ds, idxs_pit), w.ravel()))
    assert _hip.transfer_stats()["d2h_bytes"] == 0  # (reset)
