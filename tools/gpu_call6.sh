#!/bin/bash
mkdir -p gpurun_out/r2f
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|^FAILED|^ERROR|Error|assert " gpurun_out/r2f/pytest.log | head -30
timeout 600 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; echo "bench rc $?"
tail -c 1000 gpurun_out/r2f/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r2f/bench.json'))
    for k in ('value','ms_per_step','stage_ms','per_frame_counts','pose_translation_error_m','roofline','roofline_nets'):
        print(k, json.dumps(d.get(k)))
    print(json.dumps(d.get('extra',{}).get('configs1_frontend_batched')))
    print(d.get('extra_error'), d.get('roofline_error'))
except Exception as e: print("no bench json", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2f/prof -o e2e -- python /root/repo/bench.py --no-extra --cpu-baseline 0 --steps 10 > /root/repo/gpurun_out/r2f/bench_prof.json 2> /root/repo/gpurun_out/r2f/bench_prof.err; echo "prof rc $?"
cd /root/repo
find gpurun_out/r2f/prof -name "*kernel_trace.csv" -delete; find gpurun_out/r2f/prof -name "*agent_info.csv" -delete; ls -R gpurun_out/r2f/prof | head
