#!/bin/bash
# round 3, GPU call C: detector graph fix, PnP refit; whole suite, bench (device / host hand-over), detector timeline
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_pnp_gpu.py -q -x --durations=10 > $OUT/pytest_new.txt 2>&1; echo "pytest_new rc $?" >> $OUT/pytest_new.txt
tail -25 $OUT/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -8 $OUT/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 --handover host > $OUT/bench_host.json 2> $OUT/bench_host.err; echo "bench host rc $?"
VIDO_NO_DET_GRAPH=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_nodetgraph.json 2> $OUT/bench_nodetgraph.err; echo "bench nodet rc $?"
python - <<'PY'
import json
for f in ("bench.json", "bench_host.json", "bench_nodetgraph.json"):
    try:
        d = json.load(open("gpurun_out/r3c/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["per_frame_counts"], d["config"]["net_optimisations"], d["pose_translation_error_m"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3c/" + f.replace(".json", ".err")).read()[-3000:])
PY
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tl.log 2>&1
cd $REPO
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 40 > $OUT/tl_summary.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
tail -3 $OUT/tl.log
