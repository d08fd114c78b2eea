#!/bin/bash
set -u
OUT=gpurun_out/r5p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s 2>&1 | tail -15 | tee $OUT/pytest_fullsize.txt
timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/err.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5p/bench.json")); print(d["value"], d["ms_per_step"]); print(d["stage_ms"]); print(d.get("roofline_nets")); print(d["config"].get("net_optimisations"))
PY
