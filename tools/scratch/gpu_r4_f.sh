#!/bin/bash
# round 4, GPU call F: is the collapse of the three-chain schedules the runtime's limit of 4 hardware queues per process?  (GPU_MAX_HW_QUEUES)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4f; mkdir -p $OUT
for q in 4 8 16; do
 for sch in "flow+depth" "flow,depth" "det" "det,flow+depth" "det,flow,depth"; do
  tag=$(echo "$sch" | tr '+,!' 'pcx')
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 --net-streams "$sch" > $OUT/b_${q}_$tag.json 2> $OUT/b_${q}_$tag.err
  python -c "
import json
try:
  d=json.load(open('$OUT/b_${q}_$tag.json')); print('hwq $q', '$sch', d['value'], d['ms_per_step'], 'tracker', round(d['stage_ms']['tracker_thread_ms'],2), 'wait', round(d['stage_ms']['tracker_wait_for_nets_ms'],2))
except Exception as e: print('hwq $q', '$sch', 'ERR', e)"
 done
done
