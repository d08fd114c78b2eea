"""One shape of the split-bf16 conv1x1 a few times (for rocprofv3): python tools/r6/c1b3_run.py cin cout H W [reps] [res]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
cin, cout, H, W = (int(a) for a in sys.argv[1:5]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20; res = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
x = torch.randn(1, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 1, 1) / cin ** 0.5; b = torch.randn(cout, device="cuda"); r = torch.randn(1, cout, H, W, device="cuda")
wp = pack_conv1x1(w, ops.conv1x1_layout(cin, cout, H * W)).cuda()
for _ in range(reps):
    y = ops.conv1x1_bias_act(x, wp, b, r if res else None, 0.0)
torch.cuda.synchronize()
