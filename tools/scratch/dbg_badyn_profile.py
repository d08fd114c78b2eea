import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vido_slam_amd as V
ctx = V.Context(); P = V.problems
dpr = P.synth_ba_problem(n_cam=200, n_pt=20000, kind="global", track_len=10, seed=13); dpr["max_iters"] = 3
ddy = P.synth_ba_dynamic(dpr, n_obj=3, pts_per_obj=300, seed=14, max_len=8)
r = V.ba_optimize(ctx, dpr, dynamic=ddy)
print("dyn", r["iterations"], r["lm_trials"], r["ms_solve_loop"], r["ms_setup"])
