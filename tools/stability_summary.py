"""max / median per phase over the runs of tools/stability.sh (a phase = one number position of one output line)."""
import re, sys
import numpy as np
series = {}
for ln in open(sys.argv[1]):
    if ln.startswith("=="):
        tool = ln.split(":")[1].split()[0]
        continue
    key = tool + " | " + re.sub(r"[-+]?\d+\.?\d*(e[-+]?\d+)?", "#", ln.strip())[:110]
    nums = [float(x) for x in re.findall(r"(?<![\w.])[-+]?\d+\.\d+", ln)]
    if nums:
        series.setdefault(key, []).append(nums)
worst = 0.0
for key, rows in series.items():
    n = min(len(r) for r in rows)
    a = np.array([r[:n] for r in rows])
    med = np.median(a, axis=0)
    ratio = np.where(med > 0.05, a.max(axis=0) / np.maximum(med, 1e-9), 1.0)
    worst = max(worst, float(ratio.max()))
    print(f"{len(rows):3d} x [{key[:90]}] worst max/median {ratio.max():.2f} (median {med[int(ratio.argmax())]:.2f} ms, max {a.max(axis=0)[int(ratio.argmax())]:.2f} ms)")
print(f"worst max / median over all phases: {worst:.2f}")
