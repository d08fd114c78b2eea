#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 1 2; do
for cfg in "32 128 8 16 10 0" "1024 128 8 16 10 0" "2048 128 8 16 10 0" "1024 1024 8 16 10 0" "1024 128 50 68 10 0" "1024 1024 50 68 10 0" "1024 1024 50 68 10 1"; do
  rm -rf /tmp/prof; VIDO_CONV1X1_B3_GROUPS=$g timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o t -- python $R/tools/r6/c1b3_run.py $cfg > /tmp/kt.log 2>&1 || tail -3 /tmp/kt.log
  echo "== groups $g: $cfg"; python3 - $(find /tmp/prof -name "*kernel_trace.csv") <<'PY'
import csv, sys
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "conv1x1_b3" in r["Kernel_Name"]]
print(" ".join("%.1f" % x for x in d))
PY
done
done
