"""csrc/wino.hip against the library 3x3 convolution (+ the bias / activation pass it needs) on the shapes the three network nodes run at the benchmark's feeds (LiteFlowNet
at 480 x 640, the detector at 800 x 1088): microseconds per call and fp32 TFLOP/s counted as a DIRECT convolution (2 * 9 * cin * cout * positions).  Run plainly for HIP-event
timings, or under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_wino3x3
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
shapes = [(1, 131, 128, 240, 320, "flow L2 regularisation 1"), (1, 128, 128, 240, 320, "flow L2 regularisation 2"), (1, 128, 64, 240, 320, "flow L2 128->64"), (1, 64, 32, 240, 320, "flow L2 64->32"),
          (1, 49, 128, 240, 320, "flow L2 matching 1"), (2, 32, 32, 240, 320, "flow features netTwo"), (1, 130, 128, 120, 160, "flow L3 sub-pixel 1"), (1, 128, 64, 60, 80, "flow L4 128->64"),
          (1, 195, 128, 15, 20, "flow L6 regularisation 1"), (1, 128, 64, 120, 160, "flow L3 128->64"), (1, 64, 64, 120, 160, "flow L3 features"), (1, 64, 32, 120, 160, "flow L3 64->32"),
          (1, 194, 128, 60, 80, "flow L4 regularisation 1"), (1, 96, 96, 60, 80, "flow L4 features"), (1, 258, 128, 30, 40, "flow L5 regularisation 1"), (1, 128, 64, 30, 40, "flow L5 128->64"),
          (1, 386, 128, 15, 20, "flow L6 regularisation 1b"), (1, 64, 32, 15, 20, "flow L6 64->32"),
          (1, 256, 256, 200, 272, "FPN P2 / RPN P2"), (1, 256, 256, 100, 136, "FPN P3 / RPN P3"), (1, 256, 256, 50, 68, "FPN P4"), (1, 256, 256, 25, 34, "FPN P5"), (100, 256, 256, 14, 14, "mask head x100")]
def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
only = os.environ.get("WINO_ONLY")
for N, cin, cout, H, W, name in shapes:
    x = torch.randn(N, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / (3.0 * cin ** 0.5); b = torch.randn(cout, device="cuda")
    gf = 2.0 * 9 * N * cin * cout * H * W / 1e9
    up = pack_wino3x3(w).cuda()
    t1 = timed(lambda: ops.wino3x3_bias_act(x, up, b, cout, 0.1))
    line = "%-28s %d x %3d -> %3d @ %3dx%3d %6.2f GF | ours + bias + lrelu %7.1f us (%6.1f TF)" % (name, N, cin, cout, H, W, gf, t1, gf / t1 * 1e3)
    if N * cin * H * W < 40e6:                                 # the K-split form (what vido_wino3x3_form gives launches of < 128 workgroups)
        upk = pack_wino3x3(w, 1).cuda(); tk = timed(lambda: ops.wino3x3_bias_act(x, upk, b, cout, 0.1, 1)); tk2 = timed(lambda: ops.wino3x3_bias_act(x, upk, b, cout, 0.1, 2))
        line += " | K-split x4 %7.1f us, x2 %7.1f us%s" % (tk, tk2, " *" if ops.wino3x3_form(N, cin, cout, H, W) else "  ")
    if not only:
        t_lib = timed(lambda: F.conv2d(x, w, None, 1, 1)); t_lib_ep = timed(lambda: ops.bias_act_(F.conv2d(x, w, None, 1, 1), b, 0.1))
        y = ops.wino3x3_bias_act(x, up, b, cout, 0.1); ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.1)
        line += " | library %7.1f us (%6.1f TF), + bias/lrelu pass %7.1f us | max err %.2e of %.1f" % (t_lib, gf / t_lib * 1e3, t_lib_ep, float((y.double() - ref).abs().max()), float(ref.abs().max()))
    print(line, flush=True)
