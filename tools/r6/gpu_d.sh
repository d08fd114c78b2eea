#!/bin/bash
# headline A/B: 1x1 layers below 160 tiles on the split-bf16 kernel (VIDO_CONV1X1_MIN_TILES=100), and the b3 forms
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 60 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; python - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
s = d["stage_ms"]
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "det", s["maskrcnn_x101_fpn_ms"], "lfn", s["liteflownet_ms"], "trk", s["tracker_thread_ms"], "lba", s["local_ba_ms"], "nodet", d["extra"]["e2e_without_detector"]["frames_per_s"])
PY
}
run base A=1
run mt100 VIDO_CONV1X1_MIN_TILES=100
run form3 VIDO_CONV1X1_B3_FORM=3
run form1 VIDO_CONV1X1_B3_FORM=1
run f32 VIDO_CONV1X1_ARITH=f32
