#!/bin/bash
# round 4, GPU call I: lean single-frame ORB (one-launch pyramid, no stage events, blur in line), BA uploads as one copy, scalars via the last workgroup, bawin ingest: full suite + tracker + bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4i; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest.txt | tail -5
timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker.json 2> $OUT/tracker.err; cat $OUT/tracker.json
VIDO_ORB_FULL_TIMING=1 timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_fullorb.json 2>/dev/null; cat $OUT/tracker_fullorb.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tracker -o tr -- python $REPO/tools/prof_tracker.py 60 > $OUT/prof_tracker.log 2>&1
cd $REPO
VIDO_CALL_PROF=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_prof.json 2> $OUT/bench_prof.err; grep "prof\]" $OUT/bench_prof.err | head -12
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json, csv
for f in ("bench_prof.json", "bench.json"):
    try:
        d = json.load(open("gpurun_out/r4i/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r4i/" + f.replace(".json", ".err")).read()[-1500:])
rows = list(csv.DictReader(open("gpurun_out/r4i/prof_tracker/tr_kernel_stats.csv")))
print("dispatches per frame", sum(int(r["Calls"]) for r in rows) / 60.0)
PY
