"""csrc/convdirect.hip against the library convolution (+ the bias / activation pass it needs) on LiteFlowNet's layers at 480 x 640: microseconds per call and fp32 TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
shapes = [(2, 3, 32, 480, 640, (7, 7), 1, "features netOne (stem)"), (2, 32, 32, 480, 640, (3, 3), 2, "features netTwo s2"), (2, 32, 64, 240, 320, (3, 3), 2, "features netThr s2"),
          (2, 64, 96, 120, 160, (3, 3), 2, "features netFou s2"), (2, 96, 128, 60, 80, (3, 3), 2, "features netFiv s2"), (2, 128, 192, 30, 40, (3, 3), 2, "features netSix s2"),
          (1, 32, 49, 240, 320, (7, 1), 1, "L2 dist 7x1"), (1, 49, 49, 240, 320, (1, 7), 1, "L2 dist 1x7"), (1, 32, 25, 120, 160, (5, 1), 1, "L3 dist 5x1"), (1, 25, 25, 120, 160, (1, 5), 1, "L3 dist 1x5"),
          (1, 32, 25, 60, 80, (5, 1), 1, "L4 dist 5x1"), (1, 25, 25, 60, 80, (1, 5), 1, "L4 dist 1x5"), (1, 32, 9, 30, 40, (3, 3), 1, "L5 dist 3x3"), (1, 32, 9, 15, 20, (3, 3), 1, "L6 dist 3x3")]
def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
with torch.no_grad():
    for N, cin, cout, H, W, k, s, name in shapes:
        conv = torch.nn.Conv2d(cin, cout, k, s, (k[0] // 2, k[1] // 2)).cuda(); x = torch.randn(N, cin, H, W, device="cuda")
        y = ops.conv_direct_conv(conv, x, 0.1)
        gf = 2.0 * k[0] * k[1] * cin * cout * y.shape[0] * y.shape[2] * y.shape[3] / 1e9
        t1 = timed(lambda: ops.conv_direct_conv(conv, x, 0.1))
        t_lib = timed(lambda: ops.bias_act_(F.conv2d(x, conv.weight, None, s, (k[0] // 2, k[1] // 2)), conv.bias, 0.1))
        ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), s, (k[0] // 2, k[1] // 2)), 0.1)
        print("%-26s %d x %3d -> %3d @ %3dx%3d k %s s %d %6.2f GF | ours %7.1f us (%6.1f TF) | library + bias/lrelu pass %7.1f us | max err %.2e of %.1f" %
              (name, N, cin, cout, H, W, k, s, gf, t1, gf / t1 * 1e3, t_lib, float((y.double() - ref).abs().max()), float(ref.abs().max())), flush=True)
