#!/bin/bash
# round 4, call U: the enqueued-ahead local BA: parity tests, tracker alone, headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4u; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ba_gpu.py tests/test_facade_gpu.py tests/test_system_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/prof_tracker.py 60 > gpurun_out/r4u/tracker.json 2> gpurun_out/r4u/tracker.err; tail -c 900 gpurun_out/r4u/tracker.json
VIDO_BA_NO_SPEC=1 timeout 300 python tools/prof_tracker.py 60 > gpurun_out/r4u/tracker_nospec.json 2> gpurun_out/r4u/tracker_nospec.err; tail -c 900 gpurun_out/r4u/tracker_nospec.json
timeout 600 python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 > gpurun_out/r4u/bench.json 2> gpurun_out/r4u/bench.err
VIDO_BA_NO_SPEC=1 timeout 600 python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 > gpurun_out/r4u/bench_nospec.json 2> gpurun_out/r4u/bench_nospec.err
python - <<'P'
import json
for n in ("bench", "bench_nospec"):
    try:
        d = json.loads(open("gpurun_out/r4u/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("tracker_thread_ms", "tracker_wait_for_nets_ms", "update_mask_ms", "local_ba_ms", "cam_pose_ms", "obj_motion_ms")})
    except Exception as e:
        print(n, "failed", e)
P
