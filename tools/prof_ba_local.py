"""configs[3]: the 20-keyframe local window (static graph), two solves — the workload of roofline_ba; run under rocprofv3 by tools/profile_round2.sh"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
if os.environ.get('VIDO_LIB_PATH'): V.host.LIB_PATH = os.environ['VIDO_LIB_PATH']
ctx = V.Context(width=640, height=480, max_batch=1)
pr = V.problems.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
for _ in range(3): V.ba_optimize(ctx, pr)
import numpy as np
rs = [V.ba_optimize(ctx, pr) for _ in range(5)]
r = rs[0]; r["ms_solve_loop"] = float(np.median([x["ms_solve_loop"] for x in rs]))
print("iters", r["iterations"], "loop ms %.2f" % r["ms_solve_loop"])
