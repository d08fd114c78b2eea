import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
conv = torch.nn.ConvTranspose2d(256, 256, 2, 2, 0).cuda(); x = torch.relu(torch.randn(100, 256, 14, 14, device="cuda"))
def timed(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("library conv_transpose2d + bias_act: %.1f us | split-fp16 GEMM with scatter epilogue: %.1f us" % (timed(lambda: ops.bias_act_(F.conv_transpose2d(x, conv.weight, None, 2), conv.bias, 0.0)), timed(lambda: ops.deconv2x2_conv(conv, x, 0.0))))
