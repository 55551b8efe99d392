"""Sharded HAND (row blocks with seeded halo cells) on ONE GPU, blocks run one after the other: time of the
first pass (full sweep of each block), of the later passes (relaxation of the unknown cells only), exchanges needed;
against the single-handle call on the whole raster.

    python tools/bench_hand_blocks.py NROW NCOL NBLOCKS [nodata_pct] [tilt] [drain_threshold]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip, dist
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
nrow, ncol, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 30
tilt = int(sys.argv[5]) if len(sys.argv) > 5 else 100000
thr = int(sys.argv[6]) if len(sys.argv) > 6 else 100
n = nrow * ncol
kw = dict(seed=2, tilt=tilt, white=2, nodata_pct=nd)
def sync(): _hip.check(L.pfd_device_synchronize(0))
d8 = _hip.synth_d8_device(nrow, ncol, **kw)
elev = _hip.synth_elev_device(nrow, ncol, **kw)
h = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE)
upa = _hip.DeviceBuffer(n * 4)
h.upstream_area_cell(out=upa, memspace=_hip.PFD_DEVICE)
drain = _hip.DeviceBuffer(n)
band = 2000
import ctypes as C
for r0 in range(0, nrow, band):
    rows = min(band, nrow - r0)
    u = upa.download(np.int32, (rows, ncol), offset_bytes=r0 * ncol * 4)
    _hip.check(L.pfd_memcpy_h2d(0, C.c_void_p(drain.addr + r0 * ncol), _hip.ptr(np.ascontiguousarray(u > thr).view(np.uint8)), C.c_size_t(rows * ncol)))
upa.free()
whole = _hip.DeviceBuffer(n * 8)
h.hand(drain, elev, _hip.PFD_F32, out=whole, memspace=_hip.PFD_DEVICE); sync()
t0 = time.perf_counter(); h.hand(drain, elev, _hip.PFD_F32, out=whole, memspace=_hip.PFD_DEVICE); sync()
t_whole = time.perf_counter() - t0
h.close()
rows = dist.block_rows(nrow, nb)
hs, outs = [], []
for b, (r0, r1) in enumerate(rows):
    a, e = dist.block_slice(nrow, nb, b)
    hs.append(_hip.RasterHandle(d8.addr + a * ncol, r1 - r0, ncol, memspace=_hip.PFD_DEVICE, halo=dist.halo_of(b, nb)))
    outs.append(_hip.DeviceBuffer((e - a) * ncol * 8))
seeds = [np.full(2 * ncol, -np.inf) for _ in range(nb)]
prev = [None] * nb
brows, nunk = [None] * nb, [None] * nb
def sweep(b, update):
    a, e = dist.block_slice(nrow, nb, b)
    t0 = time.perf_counter()
    _, brows[b], nunk[b] = hs[b].hand_block(drain.addr + a * ncol, elev.addr + a * ncol * 4, _hip.PFD_F32, seeds[b], out=outs[b],
                                            memspace=_hip.PFD_DEVICE, update=update)
    sync()
    return round((time.perf_counter() - t0) * 1e3, 2)
# the first call on a block builds what it sweeps on (the exact-order plan of the block; PFD_BLOCK_LEVELS: its level
# structure); the second one is the sweep alone
t_cold = [sweep(b, False) for b in range(nb)]
times = [[sweep(b, False) for b in range(nb)]]
prev = [s_.copy() for s_ in seeds]
for it in range(2, 65):
    if sum(nunk) == 0:
        it -= 1
        break
    for b in range(nb):
        if b > 0: seeds[b][:ncol] = brows[b - 1][1]
        if b + 1 < nb: seeds[b][ncol:] = brows[b + 1][0]
    per = []
    for b in range(nb):
        if np.array_equal(prev[b].view(np.uint64), seeds[b].view(np.uint64)):
            continue
        per.append(sweep(b, True))
        prev[b] = seeds[b].copy()
    times.append(per)
# The same protocol as an RCCL rank runs it (DistributedRaster._hand_rccl): halo seeds in a DEVICE buffer
# (pfd_set_block_io(PFD_DEVICE)), nothing but the count of unknown cells crosses PCIe inside a pass; the boundary rows move
# between the blocks' buffers OUTSIDE the timed calls (on hardware: ncclSend / ncclRecv of 2 x ncol float64 per neighbour).
dev_times = None
if os.environ.get("PFD_TOOL_DEVICE_IO", "1") == "1":
    sbuf = [_hip.DeviceBuffer(2 * ncol * 8).upload(np.full(2 * ncol, -np.inf)) for _ in range(nb)]
    for hb in hs:
        hb.set_block_io(_hip.PFD_DEVICE)
    def sweep_dev(b, update):
        a, e = dist.block_slice(nrow, nb, b)
        t0 = time.perf_counter()
        _, _, nunk[b] = hs[b].hand_block(drain.addr + a * ncol, elev.addr + a * ncol * 4, _hip.PFD_F32, sbuf[b], out=outs[b],
                                         memspace=_hip.PFD_DEVICE, update=update)
        sync()
        return round((time.perf_counter() - t0) * 1e3, 2)
    def own_row(b, last):
        top = dist.halo_of(b, nb)[0]
        r = top + (rows[b][1] - rows[b][0] - 1 if last else 0)
        return outs[b].download(np.float64, (ncol,), offset_bytes=r * ncol * 8)
    dev_times = [[sweep_dev(b, False) for b in range(nb)]]
    seen = [np.full(2 * ncol, -np.inf) for _ in range(nb)]
    for it_dev in range(2, 65):
        if sum(nunk) == 0:
            it_dev -= 1
            break
        per = []
        first_last = [(own_row(b, False), own_row(b, True)) for b in range(nb)]
        for b in range(nb):
            sd = seen[b].copy()
            if b > 0: sd[:ncol] = first_last[b - 1][1]
            if b + 1 < nb: sd[ncol:] = first_last[b + 1][0]
            if np.array_equal(sd.view(np.uint64), seen[b].view(np.uint64)):
                continue
            seen[b] = sd
            sbuf[b].upload(sd)
            per.append(sweep_dev(b, True))
        dev_times.append(per)
    for hb in hs:
        hb.set_block_io(_hip.PFD_HOST)
print(f"{nrow}x{ncol}, {nb} row blocks, drain = upa > {thr}: whole raster on one handle {t_whole*1e3:.1f} ms (warm);")
print(f"  first call per block (builds the block's plan / level structure + full sweep) {t_cold} ms")
print(f"  exchanges {it}; full sweep per block (structure cached) {times[0]} ms; later passes (unknown cells only) {times[1:]}")
print(f"  per-GPU critical path, warm ~ max full sweep {max(times[0]):.1f} ms + sum of later maxima {sum(max(t) for t in times[1:] if t):.1f} ms"
      f"; cold ~ {max(t_cold):.1f} ms + the same")
if dev_times:
    later = sum(max(t) for t in dev_times[1:] if t)
    print(f"  with the halo seeds on the device (what an RCCL rank runs; the rows move between the passes, untimed): exchanges {it_dev}; "
          f"full sweep {dev_times[0]} ms; later passes {dev_times[1:]}")
    print(f"  per-GPU critical path, device seeds ~ {max(dev_times[0]):.1f} + {later:.1f} ms = {max(dev_times[0]) + later:.1f} ms: "
          f"{t_whole * 1e3 / (max(dev_times[0]) + later):.2f}x of the one-handle call on {nb} GPUs before the cost of the exchanges themselves")
