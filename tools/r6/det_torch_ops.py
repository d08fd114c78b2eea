import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import vido_slam_amd as V
from vido_slam_amd import pipeline, synth
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640, graphs=False)
scene = synth.convoy_scene(3)
bgr = torch.as_tensor(synth.gray_to_bgr(scene.frame(0)[0]), device="cuda")
for _ in range(3): nodes._det_fn(bgr)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    nodes._det_fn(bgr); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and ("aten::" in e.key):
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for t, c, k, sh in rows[:40]: print("%8.1f us %3d x %-28s %s" % (t, c, k, sh))
