"""variance of the global BA's host set-up: 8 solves alone in a fresh process, then the same after importing torch and running a CPU op (OpenMP pool alive)"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vido_slam_amd as V
ctx = V.Context(width=640, height=480, max_batch=1)
gpr = V.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11); gpr["max_iters"] = 5
def runs(tag, n=6):
    out = []
    for _ in range(n):
        t = time.perf_counter(); r = V.ba_optimize(ctx, gpr); w = (time.perf_counter() - t) * 1e3
        out.append((round(r["ms_setup"], 2), round(r["ms_solve_loop"], 2), round(w, 2)))
    print(tag, "(setup, loop, wall) ms:", out, flush=True)
runs("fresh process")
import torch
a = torch.randn(2048, 2048); b = a @ a
runs("after a torch CPU matmul (OpenMP pool alive)")
pass
pass
torch.set_num_threads(1)
runs("torch.set_num_threads(1)")
