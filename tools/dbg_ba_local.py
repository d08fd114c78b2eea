"""configs[3]: the 20-keyframe local window (static graph), two solves — the workload of roofline_ba; run under rocprofv3 by tools/profile_round2.sh"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
pr = V.problems.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
V.ba_optimize(ctx, pr)
r = V.ba_optimize(ctx, pr)
print("iters", r["iterations"], "loop ms %.2f" % r["ms_solve_loop"])
