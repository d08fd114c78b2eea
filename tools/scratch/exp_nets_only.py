"""Throughput of the three network nodes alone (no tracker), as pipeline.NetNodes runs them: three streams + hipGraphs, and serially on one stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd import pipeline
for streams in (True, False):
    ctx = V.Context(width=640, height=480, max_batch=1)
    nodes = pipeline.NetNodes(ctx, 480, 640, streams=streams)
    a = torch.randint(0, 255, (480, 640, 3), dtype=torch.uint8, device="cuda"); b = torch.randint(0, 255, (480, 640, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        out = nodes.infer(a, b)
    torch.cuda.synchronize()
    n = 30; t = time.perf_counter()
    for _ in range(n):
        out = nodes.infer(a, b)
    torch.cuda.synchronize()
    print("streams=%s: %.2f ms per frame (nets only, host enqueue + GPU)" % (streams, (time.perf_counter() - t) / n * 1e3))
    del nodes
