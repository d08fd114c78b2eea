"""Round 6: the split-fp16 FC kernel (csrc/fch.hip) against torch's fp32 addmm and float64: error, us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rows, k, outs, name in ((1000, 12544, 1024, "fc6"), (1000, 1024, 1024, "fc7"), (100, 12544, 1024, "fc6 at 100 rows"), (37, 64, 128, "tiny"), (1000, 12544 // 2, 256, "half")):
    S = ops.ctx.lib.vido_fc_h_splitk(rows, k, outs)
    g = torch.Generator().manual_seed(k + rows)
    x = torch.relu(torch.randn(rows, k, generator=g)); w = torch.randn(outs, k, generator=g) / k ** 0.5 * torch.exp(torch.randn(outs, 1, generator=g)); b = torch.randn(outs, generator=g)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    pre = (x.double() @ w.double().t() + b.double()); sc = pre.abs().mean(0, keepdim=True)
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda(); wp = pack_conv1x1(w.reshape(outs, k, 1, 1), 3).cuda()
    yl = torch.relu(F.linear(xc, wc, bc)); tl = timed(lambda: torch.relu(F.linear(xc, wc, bc)))
    yh = ops.fc_h(xc, wp, bc, outs, 0.0); th = timed(lambda: ops.fc_h(xc, wp, bc, outs, 0.0))
    el, eh = (float(((y.cpu().double() - ref) / sc).pow(2).mean().sqrt()) for y in (yl, yh))
    gf = 2.0 * rows * k * outs / 1e9
    print("%-16s %4d x %5d -> %4d  split %d | library fp32: rms %.3e %6.1f us %6.1f TF | split-fp16: rms %.3e max %.3e %6.1f us %6.1f TF | err ratio %.2f speed-up %.2fx flag %d"
          % (name, rows, k, outs, S, el, tl, gf / tl * 1e3, eh, float((yh.cpu().double() - ref).abs().max()), th, gf / th * 1e3, eh / el, tl / th, ops.conv1x1_range_flag()), flush=True)
