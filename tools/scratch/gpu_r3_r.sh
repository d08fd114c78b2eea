#!/bin/bash
# round 3, GPU call R: register-resident P3P
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pnp_gpu.py tests/test_facade_gpu.py tests/test_system_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.txt | tail -6
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o e2e -- python $REPO/bench.py --steps 30 --warmup 3 --cpu-baseline 0 --no-extra > $OUT/bench_prof.json 2> $OUT/bench_prof.err; cd $REPO
python - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/r3r/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_pnp", "k_pose_opt", "k_bawin")): print(r["Name"][:60], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1000))
d = json.load(open("gpurun_out/r3r/bench_prof.json")); print(d["value"], d["stage_ms"])
PY
