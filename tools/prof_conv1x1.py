"""csrc/conv1x1.hip against the library 1x1 convolution (+ the bias / residual / ReLU pass it needs) on the detector's bottleneck shapes at the 800 x 1088 feed: microseconds
per call and fp32 TFLOP/s.  Run plainly for HIP-event timings, or under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
shapes = [(256, 256, 200, 272, "layer1"), (64, 256, 200, 272, "layer1 first"), (512, 512, 100, 136, "layer2"), (1024, 1024, 50, 68, "layer3 (46 of the 69)"), (2048, 2048, 25, 34, "layer4 (not taken: 850 positions)")]
def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for cin, cout, H, W, name in shapes:
    x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5; b = torch.randn(cout, device="cuda"); r = torch.randn(1, cout, H, W, device="cuda")
    gf = 2.0 * cin * cout * H * W / 1e9
    t_lib = timed(lambda: F.conv2d(x, w)); t_lib_ep = timed(lambda: ops.bias_res_act_(F.conv2d(x, w), b, r, 0.0))
    line = "%-34s %4d -> %4d @ %3dx%3d  %6.2f GF | library %6.1f us (%5.1f TF), + bias/res/relu pass %6.1f us" % (name, cin, cout, H, W, gf, t_lib, gf / t_lib * 1e3, t_lib_ep)
    if ops.conv1x1_supported(cin, cout, H * W):
        lay = ops.conv1x1_layout(cin, cout, H * W); wp = pack_conv1x1(w, lay)
        t0 = timed(lambda: ops.conv1x1_bias_act(x, wp)); t1 = timed(lambda: ops.conv1x1_bias_act(x, wp, b, r, 0.0))
        y = ops.conv1x1_bias_act(x, wp, b, r, 0.0); ref = torch.relu(F.conv2d(x, w, b) + r)
        line += " | ours %6.1f us (%5.1f TF), with epilogue %6.1f us (%5.1f TF)  max err %.2e  [layout %d]" % (t0, gf / t0 * 1e3, t1, gf / t1 * 1e3, float((y - ref).abs().max()), lay)
    print(line)
