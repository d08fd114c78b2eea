"""Does MIOpen's fused conv+bias+ReLU (torch.miopen_convolution_relu / miopen_convolution_add_relu) beat conv + the in-place k_bias_res_act pass on the detector trunk's shapes?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd import nets, pipeline      # (sets MIOPEN_USER_DB_PATH)
ctx = V.Context(width=640, height=480, max_batch=1); ops = nets.HipOps(ctx)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
shapes = [("l1", 256, 256, 200, 272), ("l2", 512, 512, 100, 136), ("l3", 1024, 1024, 50, 68), ("l4", 2048, 2048, 25, 34)]
for name, cin, mid, h, w in shapes:
    x = torch.randn(1, cin, h, w, device="cuda"); res = torch.randn(1, cin, h, w, device="cuda")
    w1 = torch.randn(mid, cin, 1, 1, device="cuda") * 0.01; b1 = torch.randn(mid, device="cuda")
    w2 = torch.randn(mid, mid // 32, 3, 3, device="cuda") * 0.01
    sep1 = lambda: ops.bias_res_act_(F.conv2d(x, w1), b1, None, 0.0)
    fus1 = lambda: torch.miopen_convolution_relu(x, w1, b1, [1, 1], [0, 0], [1, 1], 1)
    sep2 = lambda: ops.bias_res_act_(F.conv2d(x, w2, None, 1, 1, 1, 32), b1, None, 0.0)
    fus2 = lambda: torch.miopen_convolution_relu(x, w2, b1, [1, 1], [1, 1], [1, 1], 32)
    sep3 = lambda: ops.bias_res_act_(F.conv2d(x, w1), b1, res, 0.0)
    fus3 = lambda: torch.miopen_convolution_add_relu(x, w1, res, 1.0, b1, [1, 1], [0, 0], [1, 1], 1)
    row = [name]
    for tag, a, b in (("1x1+relu", sep1, fus1), ("g3x3+relu", sep2, fus2), ("1x1+res+relu", sep3, fus3)):
        try:
            ta = timed(a)
        except Exception as e:
            ta = float("nan"); print("sep", tag, "failed", e)
        try:
            tb = timed(b); err = float((a() - b()).abs().max())
        except Exception as e:
            tb = float("nan"); err = -1; print("fused", tag, "failed:", str(e)[:200])
        row.append("%s sep %.1f us fused %.1f us err %.2g" % (tag, ta, tb, err))
    print(" | ".join(row), flush=True)
