"""Seeded up-sweeps over row blocks on ONE GPU, blocks one after the other: float32 accuflux and the
Strahler order; first call per block (builds the block's exact-order plan), rounds, sweep times; against the whole
raster on one handle.

    python tools/bench_up_blocks.py NROW NCOL NBLOCKS [nodata_pct] [tilt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflwdir_amd import _hip, dist
if os.environ.get("PFD_TOOL_RESERVE_GIB"):  # one arena for the working buffers (pfd_reserve): no hipMalloc while timing
    from pyflwdir_amd import _hip as _h0
    _h0.reserve(int(float(os.environ["PFD_TOOL_RESERVE_GIB"]) * 2**30))
L = _hip.lib()
nrow, ncol, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tilt = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 26
kw = dict(seed=0, tilt=tilt, white=2, nodata_pct=nd)
def sync(): _hip.check(L.pfd_device_synchronize(0))
d8 = _hip.synth_d8_device(nrow, ncol, **kw)
rows = (np.cos(np.linspace(-0.9, 0.9, nrow)) * 0.81).astype(np.float32)
h = _hip.RasterHandle(d8, nrow, ncol, memspace=_hip.PFD_DEVICE)
out = _hip.DeviceBuffer(nrow * ncol * 4)
t = []
for _ in range(3):
    sync(); t0 = time.perf_counter()
    h.accuflux_rows(rows, _hip.PFD_F32, nodata_i=-9999, nodata_f=-9999.0, has_nodata=1, direction=_hip.PFD_UP, mask_invalid=0, out=out, memspace=_hip.PFD_DEVICE)
    sync(); t.append(round(1e3 * (time.perf_counter() - t0), 2))
print(f"{nrow}x{ncol}: accuflux of one float32 value per row on one handle: {t} ms (first call builds the plan)")
h.close(); out.free()
for kind, dtype in (("accuflux", np.float32), ("strahler", np.uint8)):
    blocks, cold = [], []
    for b, (r0, r1) in enumerate(dist.block_rows(nrow, nb)):
        a, e = dist.block_slice(nrow, nb, b)
        hh = _hip.RasterHandle(d8.addr + a * ncol, r1 - r0, ncol, memspace=_hip.PFD_DEVICE, halo=dist.halo_of(b, nb))
        blocks.append(dist._UpBlock(hh, kind, dtype, payload=rows[a:e] if kind == "accuflux" else None, by_row=True, nodata=(-9999, -9999.0, 1)))
    if os.environ.get("PFD_UP_FULL"):  # A/B: every round sweeps the whole block (the behaviour before round 4)
        for blk in blocks: blk.incremental = False
    seeds = [np.zeros(2 * ncol, dtype) for _ in range(nb)]
    times = []
    for it in range(64):
        per = []
        for b, blk in enumerate(blocks):
            sync(); t0 = time.perf_counter()
            if blk.sweep(seeds[b]):
                sync(); per.append(round(1e3 * (time.perf_counter() - t0), 2))
        if not per:
            break
        times.append(per)
        for b in range(nb):
            if b > 0: seeds[b][:ncol] = blocks[b - 1].brows[1]
            if b + 1 < nb: seeds[b][ncol:] = blocks[b + 1].brows[0]
    bad = sum(blk.verify(seeds[b]) for b, blk in enumerate(blocks))
    print(f"  {kind} over {nb} row blocks: rounds {len(times)}, per round and block {times} ms; cells failing their local equation: {bad}")
    for blk in blocks: blk.close()
