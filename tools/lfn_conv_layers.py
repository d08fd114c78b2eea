"""Enumerates the convolution layers of LiteFlowNet at 640x480 (forward hooks on the CPU module) and the kernel each one takes in the GPU build: csrc/wino.hip where it has a
form that fills the chip, csrc/convsmall.hip for the 1x1 / 2-channel layers, the library (MIOpen / rocBLAS through torch) for the rest -> profiles/r5/lfn_conv_layers.txt"""
import sys, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vido_slam_amd as V
from vido_slam_amd import nets
from vido_slam_amd.host import load_library
from vido_slam_amd.nets.ops import correlation_torch_reference
lib = load_library()
net = nets.LiteFlowNet(correlation_torch_reference).eval()
rows = []
def hook(m, inp, out):
    x = inp[0]
    rows.append((m.in_channels, m.out_channels, tuple(m.kernel_size), tuple(m.stride), x.shape[0], x.shape[2], x.shape[3]))
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d): m.register_forward_hook(hook)
with torch.no_grad():
    net(torch.rand(1, 3, 480, 640), torch.rand(1, 3, 480, 640))
tot = 0
from collections import Counter
agg = Counter()
for cin, cout, k, s, N, H, W in rows:
    Ho, Wo = (H + s[0] - 1) // s[0], (W + s[1] - 1) // s[1]
    fl = 2.0 * N * cin * cout * k[0] * k[1] * Ho * Wo
    path = "lib"
    if k == (3, 3) and s == (1, 1) and lib.vido_wino3x3_supported(cin, cout, H, W) and lib.vido_wino3x3_fills_chip(N, cout, H, W, 0): path = "wino"
    elif k == (1, 1): path = "1x1 skinny" if cin % 2 == 0 and cin <= 256 and cout <= 256 else "lib1x1"
    elif cout == 2 and k[0] == k[1]: path = "kxk_c2"
    agg[(cin, cout, k, s, N, H, W, path)] += 1
    tot += fl
for (cin, cout, k, s, N, H, W, path), n in sorted(agg.items(), key=lambda kv: (kv[0][7], -kv[0][5] * kv[0][6])):
    Ho, Wo = (H + s[0] - 1) // s[0], (W + s[1] - 1) // s[1]
    print("%-10s x%d  %3d -> %3d  k%s s%s  N%d %3dx%3d  %7.3f GF" % (path, n, cin, cout, k, s, N, H, W, 2.0 * N * cin * cout * k[0] * k[1] * Ho * Wo / 1e9))
print("total GF", tot / 1e9)
