#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4ad; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_maskrcnn_gpu.py -x -q 2>&1 | tail -4
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 60 --warmup 8 --no-extra --cpu-baseline 0 > gpurun_out/r4ad/$tag.json 2> gpurun_out/r4ad/$tag.err; }
run base A=1
run conv3only VIDO_CONV1X1=conv3
python - <<'P'
import json
for n in ("base", "conv3only"):
    try:
        d = json.loads(open("gpurun_out/r4ad/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e:
        print(n, "failed", e)
P
