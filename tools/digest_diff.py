"""Post-processing of a tools/stress_exact.py stderr log taken with PFD_XPLAN_DIGEST=1: for every raster, the
plan-build steps whose digests differ between the two exact-engine handles."""
import sys
cur, handles, it = None, {}, None
for ln in open(sys.argv[1], errors="replace"):
    t = ln.split()
    if not t:
        continue
    if t[0] == "[handle]":
        it, cur = t[1], t[2]; handles[cur] = {}
    elif t[0] == "[xdigest]" and cur:
        handles[cur][t[1]] = t[2]
    elif t[0] == "[result]":
        a, b = handles.get("exact", {}), handles.get("exact2", {})
        diff = [k for k in a if a[k] != b.get(k)]
        for hn, hd in handles.items():
            if len({hd.get("lh"), hd.get("lh_b"), hd.get("lh_c")} - {None}) > 1:
                print("raster", t[1], hn, "k_plan_tile repeated on the same input differs:", hd.get("lh"), hd.get("lh_b"), hd.get("lh_c"))
        if diff or t[2] != "ok":
            print("raster", t[1], t[2], "differing steps:", diff)
        handles = {}
print("done")
