import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import nets
from torch.profiler import profile, ProfilerActivity
ctx = V.Context(); ops = nets.HipOps(ctx)
which = sys.argv[1] if len(sys.argv) > 1 else "lfn"
if which == "lfn":
    net = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation), 1).eval().cuda()
    a = torch.rand(1, 3, 384, 1248, device="cuda"); b = torch.rand(1, 3, 384, 1248, device="cuda")
    fn = lambda: net(a, b)
else:
    net = nets.fill_deterministic(nets.MonoDepth2(), 2).eval().cuda(); x = torch.rand(1, 3, 192, 640, device="cuda"); fn = lambda: net(x)
for _ in range(3): fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): fn()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
