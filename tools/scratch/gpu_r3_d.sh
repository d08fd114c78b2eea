#!/bin/bash
# round 3, GPU call D: experiments — MIOpen fused conv+bias+relu, linearize XCD order A/B, chain without the tracker; ROI-Align template; tests of the touched files
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3d; mkdir -p $OUT
timeout 300 python tools/scratch/exp_miopen_fused.py > $OUT/fused.txt 2>&1; tail -8 $OUT/fused.txt
for v in 0 1; do
  if [ $v = 1 ]; then export VIDO_BA_LIN_PLAIN=1; else unset VIDO_BA_LIN_PLAIN; fi
  timeout 300 python - > $OUT/lin_$v.txt 2>&1 <<'PY'
import sys, os; sys.path.insert(0, os.getcwd())
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
gpr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
V.ba_optimize(ctx, gpr)
for _ in range(3):
    r = V.ba_optimize(ctx, gpr)
    print("plain" if os.environ.get("VIDO_BA_LIN_PLAIN") else "xcd", "iters", r["iterations"], "loop ms %.3f" % r["ms_solve_loop"], "linearize us %.1f" % (r["ms_linearize_kernel"] * 1e3), "chi2 %.6f" % r["chi2_final"])
PY
  cat $OUT/lin_$v.txt | tail -3
done
unset VIDO_BA_LIN_PLAIN
timeout 600 python -m pytest tests/test_maskrcnn_gpu.py tests/test_nets_gpu.py tests/test_ba_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
VIDO_E2E_SKIP_TRACK=1 timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_notrack.json 2> $OUT/bench_notrack.err; echo "bench notrack rc $?"
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench_notrack.json", "bench.json"):
    try:
        d = json.load(open("gpurun_out/r3d/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3d/" + f.replace(".json", ".err")).read()[-2000:])
PY
