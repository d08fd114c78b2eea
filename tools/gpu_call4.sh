#!/bin/bash
mkdir -p gpurun_out/r2d
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|^FAILED|^ERROR|Error|assert " gpurun_out/r2d/pytest.log | head -40
timeout 600 python bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r2d/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r2d/bench.json'))
    for k in ('value','ms_per_step','stage_ms','per_frame_counts','pose_translation_error_m','roofline','roofline_nets','cpu_baseline'):
        print(k, json.dumps(d.get(k)))
    print(json.dumps(d['config']['net_optimisations']))
    print(json.dumps(d.get('extra',{}).get('configs1_frontend_batched')))
    print(d.get('extra_error'), d.get('roofline_error'))
except Exception as e: print("no bench json", e)
PY
