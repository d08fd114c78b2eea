#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4z; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_maskrcnn_gpu.py -x -q 2>&1 | tail -4
timeout 300 python tools/prof_conv1x1.py 2>/dev/null | cut -c1-260 | tee gpurun_out/r4z/conv1x1.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 60 --warmup 8 --no-extra --cpu-baseline 0 > gpurun_out/r4z/$tag.json 2> gpurun_out/r4z/$tag.err; }
run base A=1
run conv3only VIDO_CONV1X1=conv3
python - <<'P'
import json
for n in ("base", "conv3only"):
    try:
        d = json.loads(open("gpurun_out/r4z/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e:
        print(n, "failed", e)
P
