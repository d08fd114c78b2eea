#!/bin/bash
# headline A/B (60 steps each)
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 60 --no-extra --cpu-baseline 0 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; python - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
s = d["stage_ms"]
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "det", s["maskrcnn_x101_fpn_ms"], "lfn", s["liteflownet_ms"], "trk", s["tracker_thread_ms"], "lba", s["local_ba_ms"], "wait_nets", s["tracker_wait_for_nets_ms"])
PY
}
run gconv_b3 A=1
run gconv_f32 VIDO_NO_GCONV_B3=1
run gconv_b3_2 A=1
run gconv_f32_2 VIDO_NO_GCONV_B3=1
