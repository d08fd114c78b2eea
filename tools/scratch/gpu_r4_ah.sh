#!/bin/bash
# stride-2 grouped convolution: parity, headline A/B (VIDO_NO_GCONV_S2=1 keeps the library's strided convolution)
mkdir -p gpurun_out/r4ah
timeout 400 python -m pytest tests/test_maskrcnn_gpu.py -q 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > gpurun_out/r4ah/new.json 2> gpurun_out/r4ah/new.err
VIDO_NO_GCONV_S2=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > gpurun_out/r4ah/old.json 2> gpurun_out/r4ah/old.err
python - <<'PY'
import json
for n in ("new", "old"):
    try:
        d = json.loads(open("gpurun_out/r4ah/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e: print(n, "ERR", e)
PY
