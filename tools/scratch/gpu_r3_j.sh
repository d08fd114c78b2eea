#!/bin/bash
# round 3, GPU call I: device-resident local-BA window; NMS sweep rework
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_facade_gpu.py tests/test_pnp_gpu.py tests/test_ba_gpu.py tests/test_system_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -40 $OUT/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
VIDO_BA_HOST_WALK=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_walk.json 2> $OUT/bench_walk.err; echo "bench walk rc $?"
python - <<'PY'
import json
for f in ("bench.json", "bench_walk.json"):
    try:
        d = json.load(open("gpurun_out/r3j/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["pose_translation_error_m"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3j/" + f.replace(".json", ".err")).read()[-2000:])
PY
