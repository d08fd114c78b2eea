import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv3x3_h, pack_wino3x3
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
def timed(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for H, W in ((60, 80), (30, 40), (15, 20)):
    for cin, cout in ((49, 128), (130, 128), (131, 128), (128, 128), (128, 64), (64, 64), (64, 32), (32, 32), (128, 96)):
        if not ops.ctx.lib.vido_conv3x3_h_supported(1, cin, cout, H, W): continue
        x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5); b = torch.randn(cout, device="cuda")
        form = ops.wino3x3_form(1, cin, cout, H, W); u = pack_wino3x3(w, form).cuda(); wp = pack_conv3x3_h(w).cuda()
        tw = timed(lambda: ops.wino3x3_bias_act(x, u, b, cout, 0.1, form)); th = timed(lambda: ops.conv3x3_h_bias_act(x, wp, b, cout, 0.1))
        print("%3dx%3d %3d -> %3d  wgs %3d | wino form %d %6.1f us | direct %6.1f us | %.2fx" % (H, W, cin, cout, ops.ctx.lib.vido_conv3x3_h_workgroups(1, cout, H, W), form, tw, th, tw / th), flush=True)
