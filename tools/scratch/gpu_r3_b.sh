#!/bin/bash
# round 3, GPU call B: rewritten ROI-Align / NMS kernels, static detector graph, device hand-over; then the whole suite and a bench line
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_maskrcnn_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_system_gpu.py tests/test_facade_gpu.py -q -x --durations=10 > $OUT/pytest_new.txt 2>&1; echo "pytest_new rc $?" >> $OUT/pytest_new.txt
tail -25 $OUT/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -8 $OUT/pytest.txt
timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 --handover host > $OUT/bench_host.json 2> $OUT/bench_host.err; echo "bench host rc $?"
python - <<'PY'
import json
for f in ("bench.json", "bench_host.json"):
    try:
        d = json.load(open("gpurun_out/r3b/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["per_frame_counts"], d["config"]["net_optimisations"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3b/" + f.replace(".json", ".err")).read()[-3000:])
PY
