#!/bin/bash
# round 3, GPU call H: fused detector selection kernels (detpost.hip)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_maskrcnn_gpu.py tests/test_e2e_gpu.py tests/test_nets_modules_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -30 $OUT/pytest.txt
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench.json",):
    try:
        d = json.load(open("gpurun_out/r3h/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["config"]["net_optimisations"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3h/" + f.replace(".json", ".err")).read()[-2000:])
PY
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tl.log 2>&1
cd $REPO
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 45 > $OUT/tl_summary.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
