#!/bin/bash
# per-kernel durations of the local-BA configuration (20 KF x 2k landmarks): rocprofv3 kernel stats of tools/dbg_ba_profile.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lba -o lba -- python $GRAFT_REPO_ROOT/tools/dbg_ba_profile.py ${1:-local} > /tmp/lba.log 2>&1
tail -1 /tmp/lba.log
F=$(find /tmp/lba -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-46s calls %5s avg_us %8.1f min_us %8.1f total_ms %7.2f" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
