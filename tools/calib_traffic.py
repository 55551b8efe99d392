"""Known-bytes streaming kernels for the PMC calibration (tools/prof_calib.sh): 2 GiB read and written once per access
width (1, 4, 8, 16 bytes per lane) — far beyond the 256 MiB Infinity Cache, so every byte comes from / goes to HBM."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyflwdir_amd import _hip
L = _hip.lib()
L.pfd_calib_traffic.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
N = 2 << 30
buf = _hip.DeviceBuffer(N)
for w in (1, 4, 8, 16):
    _hip.check(L.pfd_calib_traffic(0, C.c_void_p(buf.addr), N, w, 1))
for w in (1, 4, 8, 16):
    _hip.check(L.pfd_calib_traffic(0, C.c_void_p(buf.addr), N, w, 0))
_hip.check(L.pfd_calib_traffic(0, C.c_void_p(buf.addr), N, 256, 0))  # one 4-byte read per 256 bytes
print("calibration kernels done:", N, "bytes each")
