#!/bin/bash
# round 3, GPU call P: MonoDepth2 decoder glue as single passes
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_nets_modules_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.txt | tail -8
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
VIDO_NO_DEPTH_FUSED=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_off.json 2> $OUT/bench_off.err; echo "bench off rc $?"
python - <<'PY'
import json
for f in ("bench.json", "bench_off.json"):
    try:
        d = json.load(open("gpurun_out/r3p/" + f)); print(f, d["value"], d["ms_per_step"], {k: v for k, v in d["stage_ms"].items() if "net" in k or "rcnn" in k or "flow" in k or "depth" in k})
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3p/" + f.replace(".json", ".err")).read()[-2000:])
PY
