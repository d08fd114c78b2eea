#!/bin/bash
# round 5, call g: bench line after the hygiene edits (default run), conv1x1 tile-form A/B on the headline
set -u
OUT=gpurun_out/r5g; mkdir -p $OUT
timeout 900 python bench.py --steps 60 --cpu-baseline 0 > $OUT/bench_default60.json 2> $OUT/bench_default60.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/bench_default60.json"))
print(d["value"], d["ms_per_step"]); print(d["roofline"]); print(d["extra"].get("e2e_without_detector")); print(d.get("roofline_ba_schur")); print(d["config"]["inputs"][:200]); print(d.get("roofline_pose_opt"))
print(d["extra"]["configs1_frontend_batched"]); print(d.get("global_ba_iters_per_s"))
PY
for rep in 1 2; do for tn in 128 0; do echo "== rep $rep VIDO_CONV1X1_TN=$tn" | tee -a $OUT/ab.txt; VIDO_CONV1X1_TN=$tn timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['maskrcnn_x101_fpn_ms'])" | tee -a $OUT/ab.txt; done; done
