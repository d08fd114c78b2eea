#!/bin/bash
# HBM traffic of the BA kernels on the local-window configuration (configs[3], 20 KF x 2k landmarks): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (MI355X_MICROARCH.md), summarised into gpurun_out/pmc_traffic_ba.json (-> profiles/<round>/).
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/bapmc/pmc_$C -- python $REPO/tools/prof_ba_pmc.py local > /tmp/bapmc_$C.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, json, os
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob("/tmp/bapmc/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[(k, row["Dispatch_Id"])] = acc.get((k, row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
        for (k, _), v in acc.items(): per.setdefault(k, []).append(v)
    for k, vs in per.items():
        res.setdefault(k, {})[c + "_KB_mean_per_launch"] = round(sum(vs) / len(vs), 1); res[k]["launches_" + c] = len(vs)
for k, d in res.items():
    d["traffic_bytes_fetch_x2"] = int((2 * d.get("FETCH_SIZE_KB_mean_per_launch", 0) + d.get("WRITE_SIZE_KB_mean_per_launch", 0)) * 1024)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"command": "tools/profile_ba_pmc.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, tools/prof_ba_pmc.py local: 20 KF x 2k landmarks, 35 581 edges)",
           "units": "counters in KB; traffic_bytes_fetch_x2 = (2*FETCH + WRITE)*1024 (gfx950 FETCH correction of MI355X_MICROARCH.md)", "kernels": res}, open("gpurun_out/pmc_traffic_ba.json", "w"), indent=1)
for k in ("k_ba_linearize", "k_ba_schur<0>", "k_ba_chol_small6", "k_ba_backsub"): print(k, res.get(k))
PY
