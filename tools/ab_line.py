"""stdin: the output of bench.py; prints ms_per_step (mean / median / min) and the phase split of its JSON line (tools/ab.sh)."""
import json
import sys

d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith("{")][-1])
ph = d.get("roofline", {}).get("phases_ms") or d.get("roofline", {}).get("phases") or {}
print(f"ms_per_step {d['ms_per_step']} median {d.get('ms_per_step_median')} min {d.get('ms_per_step_min')} phases {ph}")
