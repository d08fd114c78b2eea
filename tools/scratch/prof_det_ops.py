"""Which torch op launches which small kernel in the detector (eager _det_fn under torch.profiler): kernel name -> calling aten ops with counts and device time."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd import synth, pipeline
W, H = 640, 480
ctx = V.Context(device=0, width=W, height=H, max_batch=1)
nodes = pipeline.NetNodes(ctx, H, W)
scene = synth.convoy_scene(4, w=W, h=H, seed=5)
fr = torch.as_tensor(synth.gray_to_bgr(scene.frame(1)[0]), device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "det"
fn = (lambda: nodes._det_fn(fr)) if which == "det" else (lambda: nodes._flow_fn(fr, fr)) if hasattr(nodes, "_flow_fn") else None
for _ in range(2): fn()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fn(); torch.cuda.synchronize()
ev = prof.events()
# map kernels to their launching cpu op by correlation through time ranges: use key_averages grouped by input shape for ops with self device time
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if dt > 0: rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total self device time %.1f us" % tot)
for dt, cnt, key, shp in rows[:70]:
    print("%9.1f us %4d x  %-42s %s" % (dt, cnt, key[:42], shp))
