#!/bin/bash
# k_ba_schur_mfma time for the library variants under vido-slam_amd/variants/ (wrong results with the ablated ones: timing only)
cd /tmp; export TMPDIR=/tmp
for lib in /root/repo/vido-slam_amd/variants/*.so; do
  n=$(basename $lib .so)
  VIDO_LIB_PATH=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2m -o $n -- python /root/repo/tools/prof_ba_global.py > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("/root/repo/gpurun_out/r2m/${n}_kernel_stats.csv")):
    if "schur" in r["Name"]: print("$n", r["Name"][:30], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  rm -f /root/repo/gpurun_out/r2m/${n}_kernel_trace.csv
done
