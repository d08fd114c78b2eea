#!/bin/bash
# round 3, GPU call M: window assembly rework, one-pass bias+ReLU in the detector heads
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_facade_gpu.py tests/test_maskrcnn_gpu.py tests/test_e2e_gpu.py tests/test_system_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3m/bench.json")); print(d["value"], d["ms_per_step"], d["stage_ms"]); print(d.get("roofline_gconv")); print({k: (v["achieved"], v["gflop_per_frame"]) for k, v in d["roofline_nets"].items() if isinstance(v, dict)})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r3m/bench.err").read()[-3000:])
PY
