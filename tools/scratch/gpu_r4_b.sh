#!/bin/bash
# round 4, GPU call B: fused host-driven local-window loop (default), persistent solver as opt-in, batched update_mask, raw unprojection: tests + tracker A/B + bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest.txt | tail -5
VIDO_BA_PERSIST=1 timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_facade_gpu.py -q -x > $OUT/pytest_persist.txt 2>&1; grep -E "passed|failed|rror" $OUT/pytest_persist.txt | tail -3
timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_fused.json 2> $OUT/tracker_fused.err; cat $OUT/tracker_fused.json
VIDO_BA_NO_FUSED_LOCAL=1 timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_r3loop.json 2>/dev/null; cat $OUT/tracker_r3loop.json
for g in 32 128 256; do VIDO_BA_SCHUR0_GRID=$g timeout 300 python tools/prof_tracker.py 60 > $OUT/tracker_grid$g.json 2>/dev/null; cat $OUT/tracker_grid$g.json; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tracker -o tr -- python $REPO/tools/prof_tracker.py 60 > $OUT/prof_tracker.log 2>&1
cd $REPO
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
VIDO_BA_NO_FUSED_LOCAL=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_r3loop.json 2> $OUT/bench_r3loop.err
python - <<'PY'
import json
for f in ("bench.json", "bench_r3loop.json"):
    try:
        d = json.load(open("gpurun_out/r4b/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r4b/" + f.replace(".json", ".err")).read()[-1500:])
PY
