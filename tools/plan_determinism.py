"""Builds the exact-engine plan of the same raster repeatedly and compares the per-step digests
(PFD_XPLAN_DIGEST=1, library built with `make DEVTOOLS=1`): any difference between two builds names the first non-deterministic step."""
import os, sys, subprocess
os.environ.setdefault("PFD_ENABLE_KNOBS", "1")  # PFD_EXACT_LEVELS below is a test-only switch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    import numpy as np
    from pyflwdir_amd import _hip
    nrow, ncol, seed, tilt, nd, reps = map(int, sys.argv[2:8])
    d8 = _hip.synth_d8_device(nrow, ncol, seed=seed, tilt=tilt, white=2, nodata_pct=nd)
    L = _hip.lib()
    L.pfd_debug_xplan.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    for r in range(reps):
        h = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE)
        info = (C.c_int64 * 8)()
        sys.stderr.write(f"[build] {r}\n"); sys.stderr.flush()
        L.pfd_debug_xplan(h._h, info, None)
        # churn the allocator between builds like the stress test does
        if r % 2 == 0:
            os.environ["PFD_EXACT_LEVELS"] = "1"
            h2 = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE); h2.order_cells(); h2.close()
            os.environ.pop("PFD_EXACT_LEVELS")
        h.close()
    sys.exit(0)
cfgs = [(8213, 7912, 293, 1 << 26, 40, 24), (21926, 24936, 599, 300, 40, 10), (4364, 4704, 34, 1 << 26, 40, 40)]
for cfg in cfgs:
    env = dict(os.environ, PFD_XPLAN_DIGEST="1")
    out = subprocess.run([sys.executable, __file__, "child"] + [str(v) for v in cfg], env=env, capture_output=True, text=True)
    builds, cur = [], None
    for ln in out.stderr.splitlines():
        if ln.startswith("[build]"):
            cur = {}; builds.append(cur)
        elif ln.startswith("[xdigest]") and cur is not None:
            _, name, val = ln.split(); cur[name] = val
    ref = builds[0] if builds else {}
    diffs = [(i, [k for k in ref if b.get(k) != ref[k]]) for i, b in enumerate(builds) if b != ref]
    print(cfg, "builds", len(builds), "rc", out.returncode, "differing builds:", diffs[:6] or "none", flush=True)
    if out.returncode != 0:
        print(out.stderr[-500:])
