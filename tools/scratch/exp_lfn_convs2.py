"""GPU kernel time (not eager call time) of LiteFlowNet's small-map convolutions, per shape, via torch.profiler"""
import os, sys
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
shapes = [((1, 128, 15, 20), (64, 128, 3, 3)), ((1, 32, 15, 20), (2, 32, 3, 3)), ((1, 258, 30, 40), (128, 258, 3, 3)), ((1, 128, 30, 40), (128, 128, 3, 3)), ((1, 128, 30, 40), (64, 128, 3, 3)), ((1, 49, 30, 40), (128, 49, 3, 3)),
          ((1, 192, 8, 10), (128, 192, 3, 3)), ((1, 128, 8, 10), (64, 128, 3, 3)), ((1, 64, 8, 10), (32, 64, 3, 3)), ((1, 128, 60, 80), (128, 128, 3, 3)), ((1, 64, 60, 80), (32, 64, 3, 3)), ((1, 128, 120, 160), (64, 128, 3, 3))]
for xs, ws in shapes:
    x = torch.randn(*xs, device="cuda"); w = torch.randn(*ws, device="cuda")
    for _ in range(3): F.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): F.conv2d(x, w, None, 1, 1)
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith("aten::")]
    tot = sum(e.device_time_total for e in ev) / 10.0
    mf = 2.0 * ws[0] * ws[1] * 9 * xs[2] * xs[3] / 1e6
    print("in %-18s w %-18s: %6.1f us GPU per conv (%4.0f MFLOP, %5.1f TF/s)  kernels: %s" % (xs, ws, tot, mf, mf / tot / 1e0 * 1e-6 * 1e6 / 1e6 * 1e0, ", ".join("%s x%d" % (e.key[:28], e.count // 10) for e in ev)))
