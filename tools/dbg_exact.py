"""debug: exact engine vs oracle on a fuzz raster; prints mismatching cells with plan info"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, ctypes as C
from oracle import oracle as O
from pyflwdir_amd import _hip
import pyflwdir_amd as pyflwdir
from test_gpu_fuzz import random_d8, SHAPES
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(1000 + seed)
shape = SHAPES[seed % len(SHAPES)]
d8 = random_d8(rng, shape, p_nodata=rng.choice([0.0, 0.1, 0.4]), p_pit=rng.choice([0.002, 0.05]),
               coherent=rng.choice([0, 4, 2, 1, -1, -1]))
idxs_ds, idxs_pit, nvalid = O.from_array(d8)
seq = O.idxs_seq(idxs_ds, idxs_pit)
n = d8.size
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
flw.idxs_seq; flw.rank; flw.upstream_area()
w = rng.random(n).astype(np.float32)
w[rng.random(n) < 0.05] = -9999
got = flw.accuflux(w.reshape(shape)).ravel()
exp = O.accuflux(idxs_ds, seq, w)
L = _hip.lib()
info = (C.c_int64 * 8)()
lh = np.empty(n, np.uint8)
L.pfd_debug_xplan.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
L.pfd_debug_xplan(flw._h._h, info, lh.ctypes.data_as(C.c_void_p))
print("plan", list(info), "shape", shape, "nvalid", nvalid, "nseq", seq.size)
bad = np.flatnonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))
print("mismatches", bad.size)
rank = O.rank(idxs_ds)[0]
nup = O.upstream_count(idxs_ds)
for i in bad[:30]:
    print(i, divmod(int(i), shape[1]), "lh", lh[i], "got", got[i], "exp", exp[i], "w", w[i], "rank", rank[i], "nup", nup[i], "code", d8.flat[i])
print("lh hist", np.unique(lh, return_counts=True))
