#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/<round>/:
<round>_kernel_stats.csv (copy of the --stats kernel table) and pmc_traffic.json (mean FETCH_SIZE / WRITE_SIZE per
launch and kernel; the counters are in KB)."""
import csv, glob, json, os, shutil, sys

out, rnd = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles_" + rnd)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "bench_kernel_stats.csv"))
if os.path.exists(os.path.join(out, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(out, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof.json"))
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
            key = (k, row.get("Dispatch_Id"))
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
        per = {}
        for (k, _), v in acc.items():
            per.setdefault(k, []).append(v)
        for k, vs in per.items():
            res.setdefault(k, {})[c + "_KB_mean_per_launch"] = round(sum(vs) / len(vs), 1)
            res[k]["launches_" + c] = len(vs)
for k, d in res.items():
    d["traffic_bytes_raw"] = int((d.get("FETCH_SIZE_KB_mean_per_launch", 0) + d.get("WRITE_SIZE_KB_mean_per_launch", 0)) * 1024)
json.dump({"command": "tools/profile_round.sh " + rnd + " (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, bench.py --steps 3 --warmup 1 --no-extra, 64 frames 640x480 per launch)",
           "units": "FETCH_SIZE/WRITE_SIZE are KB; traffic_bytes_raw = (FETCH+WRITE)*1024 with no correction (the guide's x2 FETCH correction is calibrated for 16 B/lane streams only; see DESIGN.md)",
           "kernels": res}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("wrote", dst, sorted(os.listdir(dst)))
