#!/bin/bash
# round 3, GPU call S: the tracker alone (uncontended kernel durations)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_facade_gpu.py -q -x -k "resident or offline or reference_disk" > $OUT/pytest.txt 2>&1; grep -E "passed|failed" $OUT/pytest.txt | tail -2
python tools/prof_tracker.py 60 2> $OUT/t.err | tail -1
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o tr -- python $REPO/tools/prof_tracker.py 60 > $OUT/tr.json 2> $OUT/tr.err; cd $REPO
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3s/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f))); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms per frame %.3f" % (tot / 1e6 / 60))
    for r in rows[:26]: print("%-62s %5s avg us %8.1f  ms/frame %.3f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1000, float(r["TotalDurationNs"]) / 1e6 / 60))
PY
