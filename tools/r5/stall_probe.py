"""Does a small (H2D copy + tiny kernel + stream sync) on its own stream stall while LiteFlowNet's graph runs on another stream?  Isolates the 2 ms stall of the async window
solve's first operation beside the networks."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import pipeline
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640)
ex = torch.zeros((480, 640, 3), dtype=torch.uint8, device="cuda")
stop = False
def nets():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            nodes.g_flow(ex, ex); nodes.g_depth(ex)
            s.synchronize()
def probe(tag, prio, with_copy, with_kernel, n=300):
    s = torch.cuda.Stream(priority=prio)
    h = torch.zeros(30000, dtype=torch.float32).pin_memory(); d = torch.zeros(30000, dtype=torch.float32, device="cuda"); e = torch.zeros(64, device="cuda")
    lat = []
    with torch.cuda.stream(s):
        for _ in range(n):
            t0 = time.perf_counter()
            if with_copy: d.copy_(h, non_blocking=True)
            if with_kernel: e.add_(1.0)
            s.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
            time.sleep(0.002)
    lat = np.array(lat[20:])
    print("%-34s prio %2d  mean %.3f ms  p50 %.3f  p90 %.3f  max %.3f" % (tag, prio, lat.mean(), np.percentile(lat, 50), np.percentile(lat, 90), lat.max()), flush=True)
print("--- GPU idle")
for prio in (0, -1):
    probe("copy + kernel", prio, True, True); probe("kernel only", prio, False, True); probe("copy only", prio, True, False)
t = threading.Thread(target=nets); t.start(); time.sleep(0.5)
print("--- beside LiteFlowNet + MonoDepth2 graphs")
for prio in (0, -1):
    probe("copy + kernel", prio, True, True); probe("kernel only", prio, False, True); probe("copy only", prio, True, False)
stop = True; t.join()
