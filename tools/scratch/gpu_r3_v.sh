#!/bin/bash
# round 3, GPU call V: default bench with the two-chain schedule + e2e / pipeline tests
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3v; mkdir -p $OUT
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_nets_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|Error" $OUT/pytest.txt | tail -3
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout 600 python bench.py --steps 200 --warmup 10 --no-extra --cpu-baseline 0 > $OUT/bench200.json 2> $OUT/bench200.err
VIDO_E2E_SKIP_TRACK=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_notrack.json 2> $OUT/bench_notrack.err
python - <<'PY'
import json
for f in ("bench.json", "bench200.json", "bench_notrack.json"):
    try:
        d = json.load(open("gpurun_out/r3v/" + f)); print(f, d["value"], d["ms_per_step"], d["steps"], d["stage_ms"], d["per_frame_counts"]["static_points"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3v/" + f.replace(".json", ".err")).read()[-1500:])
PY
