#!/bin/bash
# round 4, GPU call E: pinned staging for every small tracker copy (one copy each way): tests + pipeline call profile + bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_track_gpu.py tests/test_system_gpu.py tests/test_facade_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest.txt | tail -5
VIDO_CALL_PROF=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_prof.json 2> $OUT/bench_prof.err; grep "prof\]" $OUT/bench_prof.err | head -40
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err
for nt in 256; do
VIDO_LISTS_NT=$nt timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_nt$nt.json 2> $OUT/bench_nt$nt.err
done
timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker.json 2> $OUT/tracker.err; cat $OUT/tracker.json
python - <<'PY'
import json
for f in ("bench_prof.json", "bench.json", "bench_nt256.json"):
    try:
        d = json.load(open("gpurun_out/r4e/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
