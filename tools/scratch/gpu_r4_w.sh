#!/bin/bash
# round 4, call W: DMA-staged grouped convolution (k_gconv3x3_m32d), conv1x1 at positions not a multiple of 4, conv1 / shortcut on our GEMM by default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4w; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_maskrcnn_gpu.py tests/test_nets_modules_gpu.py -x -q 2>&1 | tail -6
timeout 300 python tools/prof_gconv.py 2>/dev/null | tail -12 | cut -c1-220
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 60 --warmup 8 --no-extra --cpu-baseline 0 > gpurun_out/r4w/$tag.json 2> gpurun_out/r4w/$tag.err; }
run base A=1
run nodma VIDO_GCONV_NO_DMA=1
python - <<'P'
import json
for n in ("base", "nodma"):
    try:
        d = json.loads(open("gpurun_out/r4w/%s.json" % n).read().strip().splitlines()[-1]); s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e:
        print(n, "failed", e)
P
