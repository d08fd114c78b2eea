"""Distribution of the per-call time of the split-bf16 conv1x1 on one shape: python tools/r6/c1b3_spread.py cin cout H W"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 1, 1) / cin ** 0.5; b = torch.randn(cout, device="cuda"); r = torch.randn(1, cout, H, W, device="cuda")
wp = pack_conv1x1(w, ops.conv1x1_layout(cin, cout, H * W)).cuda()
fn = lambda: ops.conv1x1_bias_act(x, wp, b, r, 0.0)
fn(); fn(); torch.cuda.synchronize()
for rnd in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    ev[0].record()
    for i in range(50):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(50)]
    print("round %d: total %.1f us/call; min %.1f median %.1f max %.1f; calls > 200 us: %s" % (rnd, sum(d) / 50, min(d), sorted(d)[25], max(d), [(i, round(v)) for i, v in enumerate(d) if v > 200]))
