"""upstream_area(unit="km2") on a lat/lon grid: the exact-order form beside the opt-in fixed-point form (csrc/wide.h).

    python tools/wide_probe.py [SIZE ...]        (default 10000 30000)

Prints bench.py's two km2 lines per size (first call on a fresh handle, warm call, phases, distance between the two)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyflwdir_amd import _hip

_hip.reserve(int(float(os.environ.get("PFD_TOOL_RESERVE_GIB", "64")) * 2**30))
for size in [int(x) for x in sys.argv[1:]] or [10000, 30000]:
    for ln in bench.km2_line(size, size, bench.REGIMES["river"], f"{size}x{size} river", 5, 0):
        keep = {k: ln[k] for k in ("op", "ms_per_call", "ms_per_call_min", "first_call_on_handle_ms", "quantum_km2",
                                   "max_rel_diff_to_exact_sampled") if k in ln}
        keep["frac"], keep["frac_first_call"] = ln["roofline"]["frac"], ln["roofline_first_call"]["frac"]
        keep["phases_ms"] = ln["roofline"]["phases_ms"]
        print(json.dumps(keep), flush=True)
