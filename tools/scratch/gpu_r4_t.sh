#!/bin/bash
# round 4, call T: does a high-priority detector stream move the stretch onto the chains that have slack?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4t; export TMPDIR=/tmp
run() { tag=$1; sched=$2; shift 2; env "$@" timeout 600 python bench.py --steps 60 --warmup 8 --no-extra --cpu-baseline 0 --net-streams "$sched" > gpurun_out/r4t/$tag.json 2> gpurun_out/r4t/$tag.err; }
run base "flow+depth" A=1
run dethi "det!" A=1
run dethi_fd "det!,flow+depth" A=1
run dethi_fd_q16 "det!,flow+depth" GPU_MAX_HW_QUEUES=16
run fd_q16 "flow+depth" GPU_MAX_HW_QUEUES=16
run dethi_q16 "det!" GPU_MAX_HW_QUEUES=16
python - <<'P'
import json
for n in ("base", "dethi", "dethi_fd", "dethi_fd_q16", "fd_q16", "dethi_q16"):
    try:
        d = json.loads(open("gpurun_out/r4t/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("tracker_thread_ms", "tracker_wait_for_nets_ms", "update_mask_ms", "local_ba_ms")})
    except Exception as e:
        print(n, "failed", e)
P
