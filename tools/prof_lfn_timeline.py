"""Kernel timeline of the LiteFlowNet and MonoDepth2 graphs (run under `rocprofv3 --kernel-trace`; marker kernel = torch.erfinv; tools/summarize_timeline.py cuts the trace)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import synth, pipeline
W, H = 640, 480
ctx = V.Context(device=0, width=W, height=H, max_batch=1)
nodes = pipeline.NetNodes(ctx, H, W)
scene = synth.convoy_scene(4, w=W, h=H, seed=5)
fr = [torch.as_tensor(synth.gray_to_bgr(scene.frame(k)[0]), device="cuda") for k in range(3)]
mk = torch.rand(64, device="cuda") * 0.5
wall = []
for rep in range(8):
    torch.cuda.synchronize(); torch.erfinv(mk); torch.cuda.synchronize()
    t = time.perf_counter(); (nodes.g_flow(fr[0], fr[1]) if rep < 4 else nodes.g_depth(fr[1])); torch.cuda.synchronize(); wall.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize(); torch.erfinv(mk); torch.cuda.synchronize()
print(json.dumps({"flow_x4_then_depth_x4_ms": [round(x, 3) for x in wall], "graph_error": nodes.graph_error}))
