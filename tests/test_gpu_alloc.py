"""pfd_reserve: one arena that the library's working buffers are carved from (no hipMalloc in a steady state — its latency
for multi-GiB blocks is 0.2 ms or seconds on MI355X, tools/alloc_probe_big.py), the near-fit reuse of cached blocks, and
the allocator counters.  Run in a child process: the allocator is process-wide state."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip
res = {}
s0 = _hip.alloc_stats()
_hip.reserve(768 << 20)
d8 = O.synth_d8(2917, 3119, seed=3, tilt=1 << 26, white=2, nodata_pct=4)
exp = O.upstream_area_cell(d8)[0]
w = np.random.default_rng(0).random(d8.shape).astype(np.float32)
for it in range(3):
    flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
    res.setdefault("upa_ok", True)
    res["upa_ok"] &= bool(np.array_equal(flw.upstream_area(), exp))
    acc = flw.accuflux(w)
    so = flw.stream_order()
    if it == 0:
        acc0, so0 = acc, so
        s1 = _hip.alloc_stats()
    res["same_bits"] = bool(np.array_equal(acc.view(np.uint32), acc0.view(np.uint32)) and np.array_equal(so, so0))
    del flw
s2 = _hip.alloc_stats()
res.update(reserved=s1["reserved_bytes"], arena_blocks_first=s1["arena_blocks"] - s0["arena_blocks"],
           hipmalloc_first=s1["hipmalloc_calls"] - s0["hipmalloc_calls"],
           hipmalloc_big_later=s2["hipmalloc_calls"] - s1["hipmalloc_calls"], live_after=s2["live_blocks"],
           free_after=s2["reserved_free"])
_hip.reserve(0)  # nothing live in the arena: it goes back to the driver
res["reserved_after_release"] = _hip.alloc_stats()["reserved_bytes"]
print(json.dumps(res))
''' % ROOT


def test_reserved_arena_serves_the_working_buffers(gpu_lib):
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PFD_ENABLE_KNOBS="0"))
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["upa_ok"] and r["same_bits"]
    assert r["reserved"] == 768 << 20 and r["arena_blocks_first"] > 5  # the >= 1 MiB buffers came out of the arena
    assert r["hipmalloc_big_later"] == 0  # second and third handle: cache + arena only
    assert r["reserved_after_release"] == 0  # every arena block was given back, the arena released


CHILD_SMALL = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip
_hip.reserve(8 << 20)  # far too small for the working buffers of this raster: they overflow into the class cache / hipMalloc
d8 = O.synth_d8(2500, 2600, seed=9, tilt=1 << 26, white=2, nodata_pct=2)
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
ok = bool(np.array_equal(flw.upstream_area(), O.upstream_area_cell(d8)[0]))
so = flw.stream_order()
s = _hip.alloc_stats()
del flw
print(json.dumps(dict(ok=ok, somax=int(so.max()), hipmalloc=s["hipmalloc_calls"], arena_blocks=s["arena_blocks"], reserved=s["reserved_bytes"])))
''' % ROOT


def test_an_arena_that_is_too_small_overflows_into_the_cache(gpu_lib):
    """Requests that do not fit the reserved arena fall back to the class cache / hipMalloc: results unchanged."""
    out = subprocess.run([sys.executable, "-c", CHILD_SMALL], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["ok"] and r["somax"] >= 4 and r["reserved"] == 8 << 20
    assert r["hipmalloc"] > 5  # the GB-class buffers did not come out of the 8 MiB arena


def test_a_failed_allocation_does_not_fail_the_next_call(gpu_lib, oracle):
    """hipMalloc leaves its error sticky: reported where it happened (PFD_ENOMEM) and cleared, or the first launch check
    of the NEXT call — any call, any handle — reports "out of memory" again (seen at 90000 x 90000: `rank` failed right
    behind a `floodplains` that had run out of memory, and worked on the next try)."""
    import ctypes as C

    from pyflwdir_amd import _hip

    d8 = oracle.synth_d8(300, 200, seed=2)
    h = _hip.RasterHandle(d8, 300, 200)
    exp = h.rank()
    p = C.c_void_p()
    rc = _hip.lib().pfd_malloc(0, C.c_size_t(1 << 50), C.byref(p))  # 1 PiB
    assert rc == -4 and b"out of memory" in _hip.lib().pfd_last_error().lower()  # PFD_ENOMEM
    rc = _hip.lib().pfd_reserve(0, C.c_size_t(1 << 50))
    assert rc == -4
    np.testing.assert_array_equal(h.rank(), exp)  # (launch checks in front of its first kernel)
    h.close()
