import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyflwdir_amd import _hip
L = _hip.lib()
L.pfd_debug_xplan.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
for (nr, nc, kw) in ((30000, 30000, dict(seed=0, tilt=1 << 26, white=2, nodata_pct=0)), (36000, 72000, dict(seed=2, tilt=100000, white=2, nodata_pct=30))):
    d8 = _hip.synth_d8_device(nr, nc, **kw)
    h = _hip.RasterHandle(d8, nr, nc, device=0, memspace=_hip.PFD_DEVICE)
    info = (C.c_int64 * 8)()
    _hip.check(L.pfd_debug_xplan(h._h, info, None))
    print(nr, nc, "state", info[0], "ntrunk", info[1], f"({100*info[1]/(nr*nc):.1f}% of cells)", "nchain", info[2], "nslot", info[3], "rounds", info[4])
    h.close(); d8.free()
