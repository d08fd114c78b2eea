#!/bin/bash
# round 3, GPU call K: grouped 3x3 convolution on the matrix cores
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3k; mkdir -p $OUT
timeout 600 python -m pytest tests/test_maskrcnn_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o gc -- python $REPO/tools/prof_gconv.py > $OUT/gconv.txt 2>&1; cd $REPO; grep cpg $OUT/gconv.txt
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3k/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("gconv", "Sp3Asm", "k_bias_act", "igemm", "Conv")): print(r["Name"][:90], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1000))
PY
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench.json",):
    try:
        d = json.load(open("gpurun_out/r3k/" + f)); print(f, d["value"], d["ms_per_step"], {k: v for k, v in d["stage_ms"].items() if "net" in k or "rcnn" in k or "flow" in k or "depth" in k})
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3k/" + f.replace(".json", ".err")).read()[-2000:])
PY
