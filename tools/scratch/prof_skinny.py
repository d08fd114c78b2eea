import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1, pack_conv1x1_skinny
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for cin, cout, H, W in ((256, 256, 200, 272), (64, 256, 200, 272), (256, 128, 200, 272), (32, 64, 240, 320), (32, 128, 240, 320), (64, 128, 120, 160), (256, 256, 100, 136)):
    x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5; b = torch.randn(cout, device="cuda"); r = torch.randn(1, cout, H, W, device="cuda")
    ws = pack_conv1x1_skinny(w).cuda()
    with torch.no_grad():
        t_s = timed(lambda: ops.conv1x1_skinny(x, ws, b, cout, 0.0, r)); t_s0 = timed(lambda: ops.conv1x1_skinny(x, ws, b, cout, 0.0))
        t_l = timed(lambda: ops.bias_res_act_(F.conv2d(x, w), b, r, 0.0)); t_l0 = timed(lambda: F.conv2d(x, w))
        y = ops.conv1x1_skinny(x, ws, b, cout, 0.0, r); ref = torch.relu(F.conv2d(x.double(), w.double(), b.double()) + r.double())
        line = "%3d -> %3d @ %dx%d: skinny %.1f us (with residual %.1f) | library GEMM %.1f (+ pass %.1f)" % (cin, cout, H, W, t_s0, t_s, t_l0, t_l)
        if ops.conv1x1_supported(cin, cout, H * W):
            wp = pack_conv1x1(w).cuda(); t_c = timed(lambda: ops.conv1x1_bias_act(x, wp, b, r, 0.0)); line += " | conv1x1.hip with epilogue %.1f" % t_c
        print(line, " max err %.2e" % float((y.double() - ref).abs().max()), flush=True)
