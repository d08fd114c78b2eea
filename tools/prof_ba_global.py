import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vido_slam_amd as V
if os.environ.get('VIDO_LIB_PATH'): V.host.LIB_PATH = os.environ['VIDO_LIB_PATH']
ctx = V.Context(width=640, height=480, max_batch=1)
gpr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
V.ba_optimize(ctx, gpr)
r = V.ba_optimize(ctx, gpr)
print("iters", r["iterations"], "loop ms %.2f" % r["ms_solve_loop"], "linearize us %.1f" % (1e3 * r.get("ms_linearize_kernel", 0.0)))
