#!/bin/bash
# round 2, GPU call 2: parity tests (new FAST strips, device-side Mask R-CNN heads, batched objects), bench, kernel trace (csv)
set -x
mkdir -p gpurun_out/r2b
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2b/pytest.log
tail -40 gpurun_out/r2b/pytest.log
timeout 600 python bench.py > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo "bench rc $?"
tail -c 2000 gpurun_out/r2b/bench.err
timeout 300 python tools/exp_nets.py graphs > gpurun_out/r2b/exp_nets.jsonl 2> gpurun_out/r2b/exp_nets.err; echo "exp rc $?"
cat gpurun_out/r2b/exp_nets.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2b/prof -o e2e -- python /root/repo/bench.py --no-extra --cpu-baseline 0 --steps 10 > /root/repo/gpurun_out/r2b/bench_prof.json 2> /root/repo/gpurun_out/r2b/bench_prof.err; echo "prof rc $?"
cd /root/repo
find gpurun_out/r2b/prof -name "*.csv" | head
find gpurun_out/r2b/prof -name "*kernel_trace.csv" -delete; find gpurun_out/r2b/prof -name "*agent_info.csv" -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b/bench.json'))
for k in ('value','ms_per_step','stage_ms','per_frame_counts','pose_translation_error_m','roofline','roofline_nets'):
    print(k, json.dumps(d.get(k)))
print(json.dumps(d['config']['net_optimisations']))
print(json.dumps(d.get('extra',{}).get('configs1_frontend_batched')))
PY
