# FETCH_SIZE / WRITE_SIZE of known-bytes streaming kernels, per access width:  bash tools/prof_calib.sh [tag]
#   -> gpurun_out/<tag>/calib.csv and profiles/pmc_traffic.json["_calibration_widths"] (factor = known bytes / counter bytes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-calib}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- python $R/tools/calib_traffic.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- python $R/tools/calib_traffic.py > $O/write.log 2>&1
cd $R
python tools/rocpd_pmc_summary.py $O/fetch/f_results.db $O/write/w_results.db > $O/calib.csv
python - <<PY
import csv, json, re
known = 2 << 30
tab = {}
for r in csv.DictReader(open("$O/calib.csv")):
    if "k_calib_read_strided" in r["kernel"] and r["counter"] == "FETCH_SIZE":
        kb = float(r["avg_value"])
        tab["read_4B_per_256B"] = dict(accesses=known // 256, counter_kb=kb, counter_bytes_per_access=round(kb * 1024 / (known // 256), 2))
        continue
    m = re.search(r"k_calib_(read|write)<(.*?)>", r["kernel"])
    if not m: continue
    width = {"unsigned char": 1, "unsigned int": 4, "unsigned long": 8}.get(m.group(2), 16)
    want = "FETCH_SIZE" if m.group(1) == "read" else "WRITE_SIZE"
    if r["counter"] != want: continue
    kb = float(r["avg_value"])
    tab[f"{m.group(1)}_{width}B"] = dict(known_bytes=known, counter_kb=kb, factor=round(known / (kb * 1024), 4))
p = "profiles/pmc_traffic.json"
t = json.load(open(p))
t["_calibration_widths"] = tab
json.dump(t, open(p, "w"), indent=1, sort_keys=True)
print(json.dumps(tab, indent=1))
PY
rm -rf $O/fetch $O/write
