#!/bin/bash
# round 3, GPU call T: pose optimiser with component-major Schur blocks
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_poseopt_gpu.py tests/test_facade_gpu.py tests/test_system_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|Error" $OUT/pytest.txt | tail -3
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o tr -- python $REPO/tools/prof_tracker.py 60 > $OUT/tr.json 2> $OUT/tr.err; cd $REPO; cat $OUT/tr.json
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3t/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:6]: print("%-62s %5s avg us %8.1f  ms/frame %.3f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1000, float(r["TotalDurationNs"]) / 1e6 / 60))
PY
