"""GPU experiment: ms per forward of the three network nodes at 640x480 under the inference-time rewrites (nets/fuse.py), one variant per line of
JSON on stdout.  usage: python tools/exp_nets.py [variants...]   variants: base fold graphs find cl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vido_slam_amd as V
from vido_slam_amd import nets, pipeline

variants = sys.argv[1:] or ["base", "fold", "graphs"]
H, W = 480, 640
rng = np.random.RandomState(0)
a = torch.as_tensor((rng.rand(H, W, 3) * 255).astype(np.uint8), device="cuda"); b = torch.as_tensor((rng.rand(H, W, 3) * 255).astype(np.uint8), device="cuda")


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3), round((time.perf_counter() - t) / reps * 1e3, 3)


for v in variants:
    t0 = time.perf_counter()
    ctx = V.Context(width=W, height=H, max_batch=1)
    torch.backends.cudnn.benchmark = v == "find"
    nodes = pipeline.NetNodes(ctx, H, W, optimize=v in ("fold", "graphs", "find"), graphs=v in ("graphs", "find"), streams=True, miopen_find=v == "find")
    setup = time.perf_counter() - t0
    r = {"variant": v, "setup_s": round(setup, 1), "graph_error": nodes.graph_error}
    r["liteflownet_ms(gpu,wall)"] = timed(lambda: (nodes.g_flow or nodes._flow_fn)(a, b))
    r["monodepth2_ms"] = timed(lambda: (nodes.g_depth or nodes._depth_fn)(b))
    r["maskrcnn_trunk_ms"] = timed(lambda: (nodes.g_trunk or nodes._trunk_fn)(b))
    r["maskrcnn_full_ms"] = timed(lambda: nets.analyse_image(nodes.mask_net, b, feed=nodes.mask_feed, confidence=0.8, trunk=nodes.g_trunk), reps=3)
    def three():
        f, d, m, l, ev = nodes.infer(a, b)
        for e in ev:
            torch.cuda.current_stream().wait_event(e)
    r["three_nets_concurrent_ms"] = timed(three, reps=3)
    if v == "cl":
        pass
    print(json.dumps(r), flush=True)
    del nodes, ctx
    torch.cuda.empty_cache()
