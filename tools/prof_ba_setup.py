"""Host set-up of a global bundle adjustment (configs[4] size): wall time of vido_ba_optimize as a caller sees it, its set-up share, the LM loop; VIDO_BA_VERBOSE=1 prints the
set-up's phases, VIDO_BA_HOST_THREADS sets the host pool."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
gpr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
V.ba_optimize(ctx, gpr); V.ba_optimize(ctx, gpr)
ws, ss, ls = [], [], []
for _ in range(10):
    t0 = time.perf_counter(); r = V.ba_optimize(ctx, gpr); ws.append((time.perf_counter() - t0) * 1e3); ss.append(r["ms_setup"]); ls.append(r["ms_solve_loop"])
print("threads %s: wall ms median %.2f min %.2f | set-up %.2f | LM loop %.2f | %d iterations -> %.0f it/s (wall), %.0f (loop)" %
      (os.environ.get("VIDO_BA_HOST_THREADS", "default"), np.median(ws), min(ws), np.median(ss), np.median(ls), r["iterations"], r["iterations"] / np.median(ws) * 1e3, r["iterations"] / np.median(ls) * 1e3))
