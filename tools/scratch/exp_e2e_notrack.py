import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ["VIDO_E2E_SKIP_TRACK"] = "1"
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import pipeline, synth
from vido_slam_amd.system import System
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640)
e2e = pipeline.EndToEnd(nodes, System(), feed="given")
bgr = (np.random.rand(480, 640, 3) * 255).astype(np.uint8); d = np.ones((480, 640), np.float32); f = np.zeros((480, 640, 2), np.float32); m = np.zeros((480, 640), np.int32)
for _ in range(5): e2e.push(bgr, (d, f, m))
e2e.finish(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(30): e2e.push(bgr, (d, f, m))
e2e.finish(); torch.cuda.synchronize()
print("e2e without the tracker: %.2f ms per frame" % ((time.perf_counter() - t) / 30 * 1e3))
