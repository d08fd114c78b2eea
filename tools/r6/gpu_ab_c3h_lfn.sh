#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; cd $REPO
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>/dev/null | tail -1 > $OUT/ab_$name.json; python - <<P
import json; d=json.loads(open("$OUT/ab_$name.json").read()); s=d["stage_ms"]
print("$name", d["value"], "ms/step", d["ms_per_step"], "det", s.get("maskrcnn_x101_fpn_ms"), "lfn", s.get("liteflownet_ms"), "md2", s.get("monodepth2_ms"), "trk", s.get("tracker_thread_ms"))
P
}
run wgs190 A=1; run wgs128 VIDO_CONV3X3_H_MIN_WGS=128; run wgs96 VIDO_CONV3X3_H_MIN_WGS=96; run wgs64 VIDO_CONV3X3_H_MIN_WGS=64; run wgs190b A=1; run wgs96b VIDO_CONV3X3_H_MIN_WGS=96
