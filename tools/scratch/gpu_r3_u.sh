#!/bin/bash
# round 3, GPU call U: hardware queues
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3u; mkdir -p $OUT
run() { timeout 600 env "$1" python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 $( [ "$2" = three ] && echo --streams || echo --net-streams "$2" ) > "$OUT/b.json" 2> "$OUT/b.err"
python - "$1 $2" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r3u/b.json")); print(sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms"]["local_ba_ms"], d["stage_ms"]["track_total_ms"])
except Exception as e:
    print(sys.argv[1], "ERR", e); print(open("gpurun_out/r3u/b.err").read()[-800:])
PY
}
run GPU_MAX_HW_QUEUES=8 "flow+depth"
run GPU_MAX_HW_QUEUES=8 "flow,depth"
run GPU_MAX_HW_QUEUES=8 three
run GPU_MAX_HW_QUEUES=2 "flow+depth"
run GPU_MAX_HW_QUEUES=16 "flow+depth"
run GPU_MAX_HW_QUEUES=16 three
