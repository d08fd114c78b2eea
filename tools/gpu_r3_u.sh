#!/bin/bash
# round 3, GPU call U: stream priorities for the two chains
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3u; mkdir -p $OUT
for v in "flow+depth" "flow+depth,det!" "flow+depth!,det" "depth+flow" "flow+depth,det"; do
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 --net-streams "$v" > "$OUT/b.json" 2> "$OUT/b.err"
python - "$v" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r3u/b.json")); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["net_optimisations"]["network_streams"], d["stage_ms"]["local_ba_ms"], d["stage_ms"]["track_total_ms"])
except Exception as e:
    print(sys.argv[1], "ERR", e); print(open("gpurun_out/r3u/b.err").read()[-800:])
PY
done
