#!/bin/bash
# round 4, GPU call A: cluster pose optimiser + persistent local-window solver: parity tests, tracker alone (A/B), headline bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4a; mkdir -p $OUT
timeout 600 python -m pytest tests/test_poseopt_gpu.py tests/test_ba_gpu.py tests/test_badyn_gpu.py -q -x > $OUT/pytest1.txt 2>&1; tail -5 $OUT/pytest1.txt
timeout 600 python -m pytest tests/test_facade_gpu.py tests/test_system_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest2.txt 2>&1; tail -5 $OUT/pytest2.txt
timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_new.json 2> $OUT/tracker_new.err; cat $OUT/tracker_new.json
VIDO_BA_NO_PERSIST=1 timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_nopersist.json 2> $OUT/tracker_nopersist.err; cat $OUT/tracker_nopersist.json
for g in 8 16 64; do VIDO_BA_PERSIST_WGS=$g timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_wgs$g.json 2>/dev/null; cat $OUT/tracker_wgs$g.json; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tracker -o tr -- python $REPO/tools/prof_tracker.py 60 > $OUT/prof_tracker.log 2>&1
cd $REPO
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 1500 $OUT/bench.json
VIDO_BA_NO_PERSIST=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_nopersist.json 2> $OUT/bench_nopersist.err
python - <<'PY'
import json
for f in ("bench.json", "bench_nopersist.json"):
    try:
        d = json.load(open("gpurun_out/r4a/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r4a/" + f.replace(".json", ".err")).read()[-1500:])
PY
