import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for k, H, W in ((7, 240, 320), (5, 120, 160), (5, 60, 80), (3, 30, 40), (3, 15, 20)):
    conv = torch.nn.Conv2d(32, 2, k, 1, k // 2).cuda(); x = torch.randn(1, 32, H, W, device="cuda"); r = torch.randn(1, 2, H, W, device="cuda")
    with torch.no_grad():
        t0 = timed(lambda: ops.conv_kxk_c2(conv, x, r)); t1 = timed(lambda: ops.bias_res_act_(F.conv2d(x, conv.weight, None, 1, k // 2), conv.bias, r, 1.0))
    print("k %d %dx%d: ours %.1f us | library conv + bias/residual pass %.1f us" % (k, H, W, t0, t1), flush=True)
