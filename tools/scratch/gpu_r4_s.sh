#!/bin/bash
# round 4, call S: sensitivity of the headline to the wino fill threshold and to taking every 1x1 convolution
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 60 --warmup 8 --no-extra --cpu-baseline 0 > gpurun_out/r4s/$tag.json 2> gpurun_out/r4s/$tag.err; }
run base A=1
run wgs64 VIDO_WINO_MIN_WGS=64
run wgs256 VIDO_WINO_MIN_WGS=256
run c1all VIDO_CONV1X1=all
python - <<'P'
import json
for n in ("base", "wgs64", "wgs256", "c1all"):
    try:
        d = json.loads(open("gpurun_out/r4s/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("liteflownet_ms", "maskrcnn_x101_fpn_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e:
        print(n, "failed", e)
P
