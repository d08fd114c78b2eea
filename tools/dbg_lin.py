import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
ctx = V.Context()
pr = V.problems.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
for rep in (1,):
  os.environ['VIDO_LIN_REP'] = str(rep)
  for dbg in (0, 1, 2, 3, 4, 8, 12, 15):
    for E in (1,):
        os.environ["VIDO_LIN_DBG"] = str(dbg); os.environ["VIDO_LIN_E"] = str(E)
        for _ in range(3): r = V.ba_optimize(ctx, dict(pr))
        print("rep", rep, "dbg", dbg, "E", E, "lin_us %.1f" % (r["ms_linearize_kernel"] * 1e3 / rep), "loop_ms %.2f" % r["ms_solve_loop"], r["iterations"])
