"""Does a SECOND instance of the detector graph on a second stream raise the detector's throughput?  (a) 2 N replays of one instance back to back on one stream;
(b) N replays each of two instances on two streams at once; (c) like (b) with LiteFlowNet + MonoDepth2 replays on a third stream (the headline's mix)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd import pipeline
H, W = 480, 640
ctxA = V.Context(width=W, height=H, max_batch=1); ctxB = V.Context(width=W, height=H, max_batch=1)
A = pipeline.NetNodes(ctxA, H, W); B = pipeline.NetNodes(ctxB, H, W)
assert A.g_det is not None and B.g_det is not None, (A.graph_error, B.graph_error)
ex = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda"); ex0 = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda")
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if mode == "one":
            with torch.cuda.stream(sA): A.g_det(ex); A.g_det(ex)
        elif mode == "one+flow":
            with torch.cuda.stream(sA): A.g_det(ex); A.g_det(ex)
            with torch.cuda.stream(sC): A.g_flow(ex0, ex); A.g_depth(ex); A.g_flow(ex0, ex); A.g_depth(ex)
        elif mode == "two":
            with torch.cuda.stream(sA): A.g_det(ex)
            with torch.cuda.stream(sB): B.g_det(ex)
        elif mode == "two+flow":
            with torch.cuda.stream(sA): A.g_det(ex)
            with torch.cuda.stream(sB): B.g_det(ex)
            with torch.cuda.stream(sC): A.g_flow(ex0, ex); A.g_depth(ex); A.g_flow(ex0, ex); A.g_depth(ex)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e3
for mode in ("one", "two", "one+flow", "two+flow", "one", "two", "one+flow", "two+flow"):
    run(mode, 3)
    print("%-10s %.3f ms per frame (detector%s)" % (mode, run(mode, 20), " + LiteFlowNet + MonoDepth2" if "flow" in mode else " alone"), flush=True)
