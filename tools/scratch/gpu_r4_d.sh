#!/bin/bash
# round 4, GPU call D: call / section profile of the tracker inside the pipeline (which calls inflate), list kernels with 256 threads
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4d; mkdir -p $OUT
VIDO_CALL_PROF=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_prof.json 2> $OUT/bench_prof.err; grep "prof\]" $OUT/bench_prof.err | head -60
VIDO_CALL_PROF=1 timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_prof.json 2> $OUT/tracker_prof.err; grep "prof\]" $OUT/tracker_prof.err | head -40
for nt in 256 64; do
VIDO_LISTS_NT=$nt timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_nt$nt.json 2> $OUT/bench_nt$nt.err
done
python - <<'PY'
import json
for f in ("bench_prof.json", "bench_nt256.json", "bench_nt64.json"):
    try:
        d = json.load(open("gpurun_out/r4d/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
