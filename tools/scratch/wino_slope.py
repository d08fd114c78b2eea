"""per-chunk time and fixed overhead of k_wino3x3: one round of workgroups (fewer than CUs), input channels swept"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_wino3x3
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for cout, H, W in ((64, 256, 128), (128, 256, 128), (32, 256, 256)):
    for cin in (64, 128, 256, 512, 1024):
        x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / (3.0 * cin ** 0.5); b = torch.randn(cout, device="cuda")
        up = pack_wino3x3(w).cuda()
        t = timed(lambda: ops.wino3x3_bias_act(x, up, b, cout, 0.1))
        kc = 8 if cout >= 64 else 4
        tiles = (H // 2) * (W // 2); wgs = (tiles // (64 if kc == 8 else 128)) * ((cout + 63) // 64 if kc == 8 else cout // 32)
        print("cout %3d cin %4d %dx%d wgs %3d chunks %3d: %7.1f us  -> %.3f us per chunk" % (cout, cin, H, W, wgs, cin // kc, t, t / (cin // kc)), flush=True)
