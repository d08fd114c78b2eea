#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; cd $REPO
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>/dev/null | tail -1 > $OUT/ab_$name.json; python - <<P
import json; d=json.loads(open("$OUT/ab_$name.json").read()); s=d["stage_ms"]
print("$name", d["value"], "ms/step", d["ms_per_step"], "trk", s.get("tracker_thread_ms"), "update_mask", s.get("update_mask_ms"), "frame", s.get("frame_orb_lists_ms"), "orb", s.get("orb_ms"), "cam", s.get("cam_pose_ms"), "lba", s.get("local_ba_ms"), "wait_nets", s.get("tracker_wait_for_nets_ms"), "err", d.get("pose_translation_error_m",{}).get("mean"), "kp", d.get("per_frame_counts",{}).get("keypoints"))
P
}
for i in 1 2; do run noprefetch$i VIDO_TRACK_NO_PREFETCH=1; run prefetch$i A=1; done
