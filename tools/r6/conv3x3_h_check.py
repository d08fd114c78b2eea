"""Round 6: the direct split-fp16 3x3 kernel (csrc/conv3x3h.hip) against the fp32 Winograd kernel (csrc/wino.hip) and float64 conv2d: max-abs error of both, us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv3x3_h, pack_wino3x3
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
shapes = [(1, 128, 64, 120, 160, "LFN 128->64 level 2"), (1, 64, 64, 120, 160, "LFN 64->64 level 2"), (1, 64, 32, 120, 160, "LFN 64->32 level 2"), (2, 32, 32, 240, 320, "32->32 level 1 x2"), (1, 130, 128, 120, 160, "LFN 130->128 level 2"), (1, 16, 128, 16, 16, "one block"), (2, 32, 128, 13, 21, "small, odd sizes"), (1, 256, 256, 50, 68, "P4"), (100, 256, 256, 14, 14, "mask head"), (1, 256, 256, 100, 136, "P3"), (1, 256, 256, 200, 272, "FPN / RPN at P2")]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N, cin, cout, H, W, name in shapes:
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(N, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5); b = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    xc, bc = x.cuda(), b.cuda(); gf = 2.0 * N * cin * cout * 9 * H * W / 1e9
    form = ops.wino3x3_form(N, cin, cout, H, W); u = pack_wino3x3(w, form).cuda()
    yw = ops.wino3x3_bias_act(xc, u, bc, cout, 0.1, form); ew = float((yw.cpu().double() - ref).abs().max()); tw = timed(lambda: ops.wino3x3_bias_act(xc, u, bc, cout, 0.1, form))
    wp = pack_conv3x3_h(w).cuda()
    yh = ops.conv3x3_h_bias_act(xc, wp, bc, cout, 0.1); eh = float((yh.cpu().double() - ref).abs().max()); th = timed(lambda: ops.conv3x3_h_bias_act(xc, wp, bc, cout, 0.1))
    print("%-18s %3d x %4d -> %4d @ %3dx%3d %6.2f GF | Winograd fp32 [form %d]: err %.3e %7.1f us %6.1f TF | direct split-fp16: err %.3e %7.1f us %6.1f TF | err ratio %.2f speed-up %.2fx  flag %d"
          % (name, N, cin, cout, H, W, gf, form, ew, tw, gf / tw * 1e3, eh, th, gf / th * 1e3, eh / ew, tw / th, ops.conv1x1_range_flag()), flush=True)
