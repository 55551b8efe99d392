"""Down-sweeps over row blocks (accuflux "down", stream_distance, classic order): sweeps per block and wall time of the
fixpoint iteration with the relevance gate of pyflwdir_amd/dist.py (a block sweeps again only when a halo value it DEPENDS on
changed) against every halo value counted.

    python tools/bench_down_blocks.py [SIZE] [NBLOCKS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pyflwdir_amd import _hip
from pyflwdir_amd import dist

size = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
_hip.reserve(int(os.environ.get("PFD_TOOL_RESERVE_GIB", "60")) << 30)
data = np.ones((size, size), np.float32)
real = dist.relevant_halo
REGIMES = {"river": dict(seed=0), "rough": dict(seed=0, tilt=100000, white=2), "meander": dict(seed=0, tilt=3000, white=2)}
for regime, kw in REGIMES.items():
  buf = _hip.synth_d8_device(size, size, **kw)
  d8 = buf.download(np.uint8, (size, size))
  buf.free()
  print(f"{size} x {size}, {nb} row blocks, {regime} regime {kw}", flush=True)
  for name, fn in (("accuflux down f32", lambda: dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1), direction="down")),
                   ("stream_distance cells", lambda: dist.stream_distance_blocks(d8, nb)),
                   ("accuflux up f32 (incremental)", lambda: dist.accuflux_blocks(d8, nb, data, (-9999, -9999.0, 1)))):
      res = {}
      for mode in ("every halo value", "relevant, in flow order"):
          dist.relevant_halo = (lambda rows, halo, down: None) if mode == "every halo value" else real
          fn()  # (warm: plans, allocator)
          t0 = time.perf_counter()
          out, rounds, _ = fn()
          res[mode] = (time.perf_counter() - t0, rounds, list(dist.LAST_SWEEPS), out)
      same = res["every halo value"][3].tobytes() == res["relevant, in flow order"][3].tobytes()
      for mode, (t, rounds, sweeps, _) in res.items():
          print(f"  {name:30s} {mode:22s}: {t:6.2f} s (host arrays: upload + plan + sweeps + download), rounds {rounds}, sweeps per block {sweeps}", flush=True)
      print(f"  {'':30s} same result: {same}", flush=True)
dist.relevant_halo = real
