import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
ctx = V.Context()
kind = sys.argv[1] if len(sys.argv) > 1 else "local"
if kind == "local":
    pr = V.problems.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
else:
    pr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); pr["max_iters"] = 3
for _ in range(3 if kind == "local" else 1):
    r = V.ba_optimize(ctx, pr)
print(kind, r["iterations"], r["lm_trials"], r["ms_solve_loop"], r["ms_setup"])
