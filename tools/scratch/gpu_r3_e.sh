#!/bin/bash
# round 3, GPU call E: LiteFlowNet fused regularisation passes + pair batching A/B
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_nets_modules_gpu.py tests/test_e2e_gpu.py tests/test_ba_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
for v in fused nofused pair; do
  unset VIDO_LFN_NO_FUSED VIDO_LFN_PAIR_BATCH
  [ $v = nofused ] && export VIDO_LFN_NO_FUSED=1
  [ $v = pair ] && export VIDO_LFN_PAIR_BATCH=1
  timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "bench $v rc $?"
done
python - <<'PY'
import json
for f in ("bench_fused.json", "bench_nofused.json", "bench_pair.json"):
    try:
        d = json.load(open("gpurun_out/r3e/" + f)); print(f, d["value"], d["ms_per_step"], {k: v for k, v in d["stage_ms"].items() if "net" in k or "flow" in k or "mono" in k or "mask" in k})
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3e/" + f.replace(".json", ".err")).read()[-2000:])
PY
