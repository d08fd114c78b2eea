"""conv + bias + ReLU: MIOpen's fused entry (torch.miopen_convolution_relu) vs conv2d + the in-place HIP epilogue, on detector shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd import nets
ctx = V.Context(width=640, height=480, max_batch=1); ops = nets.HipOps(ctx)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, ci, co, h, w, k, g in (("res2 1x1", 256, 256, 200, 272, 1, 1), ("res2 g3x3", 256, 256, 200, 272, 3, 32), ("res4 1x1", 1024, 1024, 50, 68, 1, 1), ("res4 g3x3", 1024, 1024, 50, 68, 3, 32), ("fpn 3x3", 256, 256, 100, 136, 3, 1), ("mask 3x3 x100", 256, 256, 14, 14, 3, 1)):
    n = 100 if "x100" in name else 1
    x = torch.randn(n, ci, h, w, device="cuda"); wt = torch.randn(co, ci // g, k, k, device="cuda") * 0.05; b = torch.randn(co, device="cuda")
    def sep():
        y = F.conv2d(x, wt, None, 1, k // 2, 1, g); return ops.bias_res_act_(y, b, None, 0.0)
    try:
        fused = lambda: torch.miopen_convolution_relu(x, wt, b, [1, 1], [k // 2, k // 2], [1, 1], g)
        err = float((fused() - sep()).abs().max()); tf = timeit(fused)
    except Exception as e:
        err, tf = -1.0, -1.0; print(name, "fused failed:", str(e)[:100])
    print("%-14s conv+epilogue %7.1f us   miopen_convolution_relu %7.1f us   maxdiff %.2e" % (name, timeit(sep), tf, err))
