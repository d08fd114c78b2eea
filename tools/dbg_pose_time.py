import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
P = V.problems
ctx = V.Context(); opt = V.Optimizer(ctx)
for n in (3000, 1200):
    s = P.synth_pose_scene(n, seed=2)
    probs = {"Flow2Cam": P.pose_problem_flow2cam(s["uv_last"], s["flow"], s["depth"], s["Twl"], s["K"], s["T_init"]), "New": P.pose_problem_new(s["Xw"], s["uv_cur"], s["K"], s["T_init"])}
    for name, pr in probs.items():
        for _ in range(3): opt.pose_optimize(pr)
        t1 = time.perf_counter()
        for _ in range(10): r = opt.pose_optimize(pr)
        print(name, n, "ms %.3f" % ((time.perf_counter() - t1) * 100), "iters", r["lm_iterations"], "inl", r["n_inliers"], "chi2 %.6f" % r["chi2_final"])
