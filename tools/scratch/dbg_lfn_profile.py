"""Per-kernel profile of the flow node as NetNodes runs it (fused epilogues, HIP warp / correlation), eager, 640x480."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vido_slam_amd as V
from vido_slam_amd import nets
ctx = V.Context(width=640, height=480, max_batch=1); ops = nets.HipOps(ctx)
net = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_, warp=ops.backwarp), 1).eval().cuda()
a = torch.randint(0, 255, (480, 640, 3), dtype=torch.uint8, device="cuda"); b = torch.randint(0, 255, (480, 640, 3), dtype=torch.uint8, device="cuda")
with torch.no_grad():
    for _ in range(3): nets.analyse_flow(net, a, b)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        nets.analyse_flow(net, a, b); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
