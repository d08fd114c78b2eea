#!/bin/bash
# round 3, GPU call F: W/Cp slot records (256 B, line-aligned); LiteFlowNet deconv kernel + residual fold; bench
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_nets_modules_gpu.py tests/test_ba_gpu.py tests/test_badyn_gpu.py tests/test_facade_gpu.py tests/test_e2e_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
timeout 300 python - > $OUT/lin.txt 2>&1 <<'PY'
import sys, os; sys.path.insert(0, os.getcwd())
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
gpr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
V.ba_optimize(ctx, gpr)
for _ in range(3):
    r = V.ba_optimize(ctx, gpr)
    print("records iters", r["iterations"], "loop ms %.3f" % r["ms_solve_loop"], "linearize us %.1f" % (r["ms_linearize_kernel"] * 1e3), "chi2 %.6f" % r["chi2_final"])
pr = V.problems.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
V.ba_optimize(ctx, pr)
for _ in range(3):
    r = V.ba_optimize(ctx, pr)
    print("local iters", r["iterations"], "loop ms %.3f" % r["ms_solve_loop"], "setup %.3f" % r["ms_setup"], "linearize us %.1f" % (r["ms_linearize_kernel"] * 1e3))
PY
cat $OUT/lin.txt | tail -6
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench.json",):
    try:
        d = json.load(open("gpurun_out/r3f/" + f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(f, "ERR", e); print(open("gpurun_out/r3f/" + f.replace(".json", ".err")).read()[-2000:])
PY
