"""Every convolution LiteFlowNet runs at 640x480 (hooked), with MIOpen's time per shape: where do the 5.4 ms go by pyramid level?"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd import nets
ctx = V.Context(width=640, height=480, max_batch=1); ops = nets.HipOps(ctx)
net = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, warp=ops.backwarp), 1).eval().cuda()      # no fused epilogue: plain modules so that hooks see every conv
shapes = []
def hook(m, inp, out):
    x = inp[0]; shapes.append((type(m).__name__, tuple(x.shape), tuple(m.weight.shape), m.stride, m.padding, m.groups, tuple(out.shape)))
for m in net.modules():
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)): m.register_forward_hook(hook)
a = torch.rand(1, 3, 480, 640, device="cuda")
with torch.no_grad(): net(a, a)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
cache = {}; by_px = collections.OrderedDict()
for s in shapes:
    if s not in cache:
        kind, xs, ws, st, pd, g, os_ = s
        x = torch.randn(*xs, device="cuda"); w = torch.randn(*ws, device="cuda")
        cache[s] = timeit((lambda: F.conv2d(x, w, None, st, pd, 1, g)) if kind == "Conv2d" else (lambda: F.conv_transpose2d(x, w, None, st, pd, 0, g)))
    px = s[6][2] * s[6][3]
    d = by_px.setdefault(px, [0, 0.0, 0.0]); d[0] += 1; d[1] += cache[s]
    kind, xs, ws, st, pd, g, os_ = s
    d[2] += 2.0 * os_[1] * os_[2] * os_[3] * ws[1] * ws[2] * ws[3] / 1e6 if kind == "Conv2d" else 0
tot = sum(v[1] for v in by_px.values())
for px, (n, us, mf) in sorted(by_px.items()):
    print("output pixels %7d: %3d convs %8.1f us  (%.1f us each, %.0f MFLOP total)" % (px, n, us, us / n, mf))
print("all convs: %d, %.2f ms" % (len(shapes), tot / 1e3))
small = [s for s in shapes if s[6][2] * s[6][3] <= 1300]
for s in sorted(set(small), key=lambda s: -cache[s])[:12]:
    print("  %s in %s w %s stride %s -> %.1f us x%d" % (s[0], s[1], s[2], s[3], cache[s], small.count(s)))
