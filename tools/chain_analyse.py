"""Dependency distances of the chain layout (debug)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from pyflwdir_amd import _hip
from oracle import oracle as O
L = _hip.lib()
n0 = int(sys.argv[1]); tilt = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 26
d8 = O.synth_d8(n0, n0, seed=0, tilt=tilt, white=2, nodata_pct=0)
h = _hip.RasterHandle(d8, n0, n0)
n = n0 * n0
pos = np.empty(n, np.uint32); seq = np.empty(n, np.uint32); nc = C.c_int64(0)
L.pfd_debug_chain_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
_hip.check(L.pfd_debug_chain_layout(h._h, _hip.ptr(pos), _hip.ptr(seq), C.byref(nc)))
print("n_chain", nc.value, "of", n)
idxs_ds, pits, _ = O.from_array(d8)
valid = idxs_ds >= 0
child = np.flatnonzero(valid & (idxs_ds != np.arange(n)))
parent = idxs_ds[child]
dpos = pos[parent].astype(np.int64) - pos[child].astype(np.int64)
print("negative deps:", int((dpos <= 0).sum()))
du = (pos[parent] >> 6).astype(np.int64) - (pos[child] >> 6).astype(np.int64)
ext = du > 0
print("edges", child.size, "external", int(ext.sum()))
hist = np.bincount(np.floor(np.log2(du[ext])).astype(int), minlength=24)
print("log2(unit distance) histogram of external edges:", hist.tolist())
upa = O.upstream_area_cell(d8)[0].ravel()
# first positions of each bucket
b = np.floor(np.log2(upa[seq[:nc.value]].clip(1))).astype(int)
print("cells whose own uparea bucket by layout decile:", [int(np.median(b[int(q*nc.value/10):int((q+1)*nc.value/10)])) for q in range(10)])
