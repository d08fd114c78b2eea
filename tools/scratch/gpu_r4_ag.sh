#!/bin/bash
# m16d (8 / 16 channels per group by buffer-load-to-LDS copies): parity + headline A/B against the round's earlier kernel (VIDO_GCONV_NO_DMA16=1)
mkdir -p gpurun_out/r4ag
timeout 300 python -m pytest tests/test_maskrcnn_gpu.py -q 2>&1 | tail -3
timeout 300 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > gpurun_out/r4ag/new.json 2> gpurun_out/r4ag/new.err
VIDO_GCONV_NO_DMA16=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > gpurun_out/r4ag/old.json 2> gpurun_out/r4ag/old.err
python - <<'PY'
import json
for n in ("new", "old"):
    try:
        d = json.loads(open("gpurun_out/r4ag/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e: print(n, "ERR", e)
PY
