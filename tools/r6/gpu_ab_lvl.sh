#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; cd $REPO
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>/dev/null | tail -1 > $OUT/ab_$name.json; python - <<P
import json; d=json.loads(open("$OUT/ab_$name.json").read()); s=d["stage_ms"]
print("$name", d["value"], "ms/step", d["ms_per_step"], "det", s.get("maskrcnn_x101_fpn_ms"), "wait_nets", s.get("tracker_wait_for_nets_ms"))
P
}
for i in 1 2; do run torch_$i VIDO_NO_ROI_LEVELS=1; run kernel_$i A=1; done
