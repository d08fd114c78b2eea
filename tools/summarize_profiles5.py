#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round5.sh into gpurun_out/profiles_r5/ (copied to profiles/r5/ and committed): the --stats kernel tables and, per
workload, the mean FETCH_SIZE / WRITE_SIZE per launch and kernel (counters are in KB; traffic_bytes_fetch_x2 = (2 FETCH + WRITE) * 1024, the gfx950 correction of
MI355X_MICROARCH.md).  bench.py reads pmc_traffic.json, pmc_traffic_ba_global.json and fast_sq_counters.txt of the newest profiles/rN/ that has them."""
import csv, glob, json, os, shutil, sys

out = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "5"                      # tools/profile_round6.sh passes 6
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles_r" + rnd)
os.makedirs(dst, exist_ok=True)
for tag, name in (("e2e", "e2e_kernel_stats.csv"), ("fe", "frontend_kernel_stats.csv"), ("bag", "global_ba_kernel_stats.csv"), ("nd", "nodet_kernel_stats.csv")):
    for f in glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True):
        rows = open(f).read().splitlines()[:60]
        open(os.path.join(dst, name), "w").write("\n".join(rows) + "\n")
for f in ("bench_under_rocprof.json", "fast_sq_counters.txt", "bench_e2e.json", "bench_e2e_200.json", "pytest_gpu.txt", "nets_mfma.json", "det_timeline_summary.txt", "nets_timeline_summary.txt",
          "conv1x1_microbench.txt", "nodet_call_profile.txt", "valu_int_issue.txt"):
    p = os.path.join(out, f)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, f))


def traffic(tag, command):
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(out, "pmc_%s_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True):
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
        d["traffic_bytes_fetch_x2"] = int((2 * d.get("FETCH_SIZE_KB_mean_per_launch", 0) + d.get("WRITE_SIZE_KB_mean_per_launch", 0)) * 1024)
    return {"command": command, "units": "FETCH_SIZE / WRITE_SIZE are KB per launch (mean over the launches of the run), collected in separate rocprofv3 --pmc passes",
            "kernels": res}


for tag, name, cmd in (("fe", "pmc_traffic.json", "tools/prof_frontend_batch.py: the batched front end, 64 frames of 640x480 per launch"),
                       ("bag", "pmc_traffic_ba_global.json", "tools/prof_ba_global.py: configs[4] size, 500 KF x 100k landmarks, 1 M edges")):
    t = traffic(tag, cmd)
    if t["kernels"]:
        json.dump(t, open(os.path.join(dst, name), "w"), indent=1)
print("wrote", dst, sorted(os.listdir(dst)))
