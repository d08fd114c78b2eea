#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out; cd $REPO
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>/dev/null | tail -1 > $OUT/ab_$name.json; python - <<P
import json; d=json.loads(open("$OUT/ab_$name.json").read()); s=d["stage_ms"]
print("$name", d["value"], "ms/step", d["ms_per_step"], "trk", s.get("tracker_thread_ms"), "lba", s.get("local_ba_ms"), "wait_nets", s.get("tracker_wait_for_nets_ms"))
P
}
run greatest A=1; run normal VIDO_CTX_PRIO=normal; run least VIDO_CTX_PRIO=least; run greatest2 A=1; run normal2 VIDO_CTX_PRIO=normal; run least2 VIDO_CTX_PRIO=least
