#!/bin/bash
# round 4, GPU call G: own 1x1-convolution GEMM: parity tests, microbenchmark against the library, detector / headline with and without it
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_maskrcnn_gpu.py -q -x -k "conv1x1 or matrix_core_1x1" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 300 python tools/prof_conv1x1.py > $OUT/microbench.txt 2> $OUT/microbench.err; cat $OUT/microbench.txt; tail -3 $OUT/microbench.err
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err
VIDO_NO_CONV1X1=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_lib.json 2> $OUT/bench_lib.err
python - <<'PY'
import json
for f in ("bench.json", "bench_lib.json"):
    try:
        d = json.load(open("gpurun_out/r4g/" + f)); print(f, d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms"].items() if k in ("maskrcnn_x101_fpn_ms", "liteflownet_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")}, d.get("roofline_nets", {}).get("maskrcnn_x101_fpn"))
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r4g/" + f.replace(".json", ".err")).read()[-1500:])
PY
