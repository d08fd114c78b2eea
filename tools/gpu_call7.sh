#!/bin/bash
# FAST strip-width variants + the BCR solver tests + bench extra
cd /root/repo
for W in 37 111 74; do
  VIDO_EXTRA_FLAGS="-DFS_MAXIW=$W" python -c "
import sys; sys.path.insert(0,'vido-slam_amd'); import build; build.build()" > /dev/null 2>&1
  echo "== FS_MAXIW=$W"; VIDO_EXTRA_FLAGS="-DFS_MAXIW=$W" timeout 120 python tools/dbg_fast_batch.py 2>&1 | tail -1
done
python -c "
import sys; sys.path.insert(0,'vido-slam_amd'); import build; build.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py tests/test_orb_gpu.py -q 2>&1 | tail -15
timeout 300 python - <<'PY'
import sys, time, os; sys.path.insert(0,'.')
import numpy as np, torch
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
P = V.problems
gpr = P.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
for env in ("1", ""):
    if env: os.environ["VIDO_BA_NO_BCR"] = "1"
    else: os.environ.pop("VIDO_BA_NO_BCR", None)
    V.ba_optimize(ctx, gpr)
    t = time.perf_counter(); r = V.ba_optimize(ctx, gpr); d = time.perf_counter() - t
    print("NO_BCR" if env else "BCR   ", "iters", r["iterations"], "trials", r["lm_trials"], "loop ms %.2f" % r["ms_solve_loop"], "chi2 %.6f -> %.6f" % (r["chi2_initial"], r["chi2_final"]), "wall %.1f" % (d * 1e3))
PY
