"""Which torch / library operations are left in the LiteFlowNet and MonoDepth2 forwards (eager, torch profiler): python tools/r6/lfn_torch_ops.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import vido_slam_amd as V
from vido_slam_amd import pipeline, synth
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640, graphs=False)
scene = synth.convoy_scene(3)
a = torch.as_tensor(synth.gray_to_bgr(scene.frame(0)[0]), device="cuda"); b = torch.as_tensor(synth.gray_to_bgr(scene.frame(1)[0]), device="cuda")
from torch.profiler import profile, ProfilerActivity
for name, fn in (("flow", lambda: nodes._flow_fn(a, b)), ("depth", lambda: nodes._depth_fn(b))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fn(); torch.cuda.synchronize()
    rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:100]) for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and "aten::" in e.key]
    rows.sort(reverse=True)
    print("==", name, "aten ops total %.0f us" % sum(r[0] for r in rows))
    for t, c, k, sh in rows[:16]: print("%8.1f us %3d x %-30s %s" % (t, c, k, sh))
