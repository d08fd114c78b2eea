"""Grouped 3x3 convolution of the detector's bottlenecks: csrc/gconv.hip against the library convolution + bias pass, HIP-event timed (us per call, 200 calls each)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
import vido_slam_amd as vido
from vido_slam_amd.nets.ops import HipOps, pack_gconv3x3

ctx = vido.Context()
ops = HipOps(ctx)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for cpg, H, W in ((8, 200, 272), (16, 100, 136), (32, 50, 68), (64, 25, 34)):
    G = 32; C = G * cpg
    x = torch.randn(1, C, H, W, device="cuda"); w = torch.randn(C, cpg, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    wp = pack_gconv3x3(w, G)
    t_lib = timeit(lambda: ops.bias_res_act_(F.conv2d(x, w, None, 1, 1, 1, G), b, None, 0.0))
    t_conv = timeit(lambda: F.conv2d(x, w, None, 1, 1, 1, G))
    t_new = timeit(lambda: ops.gconv3x3_bias_act(x, wp, b, G, 0.0))
    t_ib = timeit(lambda: ops.gconv3x3_bias_act(x, wp, b, G, 0.0, in_bias=b))
    err = float((ops.gconv3x3_bias_act(x, wp, b, G, 0.0) - ops.bias_res_act_(F.conv2d(x, w, None, 1, 1, 1, G), b, None, 0.0)).abs().max())
    fl = 2.0 * C * cpg * 9 * H * W
    print(f"cpg {cpg:2d} {H}x{W}: library conv {t_conv:6.1f} us (+bias pass {t_lib:6.1f}) | gconv.hip {t_new:6.1f} us (with input bias {t_ib:6.1f}) = {fl / t_new / 1e6:6.1f} TFLOP/s | max diff {err:.2e}", flush=True)
