#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bdy -o bdy -- python $GRAFT_REPO_ROOT/tools/dbg_badyn_profile.py > /tmp/bdy.log 2>&1
tail -2 /tmp/bdy.log
F=$(find /tmp/bdy -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-46s calls %5s avg_us %8.1f total_ms %7.2f" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
