"""Where the detector's 14.5 ms go: trunk (hipGraph) vs the eager head section (GPU time by events, host time by the clock)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd import pipeline, nets
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640)
a = torch.randint(0, 255, (480, 640, 3), dtype=torch.uint8, device="cuda")
net = nodes.mask_net
def run():
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter(); e[0].record()
    feats, logits, deltas = nodes.g_trunk(a)
    e[1].record(); t1 = time.perf_counter()
    out = net.heads(feats, logits, deltas, nodes.mask_feed)
    e[2].record(); t2 = time.perf_counter()
    keep = torch.nonzero(out["scores"] > 0.8).squeeze(1)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, int(out["boxes"].shape[0]), int(len(keep))
for _ in range(5): run()
r = [run() for _ in range(10)]
import numpy as np
m = np.mean(np.array([x[:5] for x in r]), 0)
print("trunk gpu %.2f ms | heads gpu %.2f ms | host: trunk launch %.2f, heads enqueue %.2f, nonzero wait %.2f | detections %d kept %d" % (*m, r[0][5], r[0][6]))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    feats, logits, deltas = nodes.g_trunk(a); out = net.heads(feats, logits, deltas, nodes.mask_feed); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
