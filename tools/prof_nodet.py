"""The chain of BASELINE's metric text (flow + depth + track + local BA, no detector launch) as bench.py's extra.e2e_without_detector runs it: frames/s, the tracker's per-stage
ms and, with VIDO_CALL_PROF=1, the wall time of every C-ABI call the facade makes (printed by the library at exit).  Under `rocprofv3 --kernel-trace --stats`: the kernels of
that chain."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import synth, pipeline
from vido_slam_amd.system import System
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pro = 25
scene = synth.convoy_scene(pro + n + 1, w=640, h=480, seed=5)
frames = []
for k in range(pro + n):
    g, d, f, m = scene.frame(k)
    frames.append((synth.gray_to_bgr(g), np.ascontiguousarray(d, np.float32), np.ascontiguousarray(f, np.float32), np.ascontiguousarray(m, np.int32)))
tmp = tempfile.mkdtemp(prefix="vido_prof_"); cfg = os.path.join(tmp, "settings.yaml"); bench.write_settings(cfg, scene.K, 640, 480)
ctx = V.Context(width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640); nodes.skip_detector = True
slam = System(); slam.Init(cfg, System.RGBD)
e2e = pipeline.EndToEnd(nodes, slam, n_image=10 ** 6, feed="given", handover="device")
def run(lo, hi):
    for k in range(lo, hi):
        bgr, d, f, m = frames[k]; e2e.push(bgr, (d, f, m))
    e2e.finish()
run(0, pro)
torch.cuda.synchronize(); t0 = time.perf_counter()
run(pro, pro + n)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = e2e.stats[-n:]
keys = sorted({k for s in st for k in s if k.startswith("ms_")})
print(json.dumps({"frames_per_s": round(n / dt, 2), "ms_per_step": round(dt / n * 1e3, 3), "tracker_thread_ms": round(float(np.mean(e2e.t_track[-n:])), 3),
                  "tracker_wait_ms": round(float(np.mean(e2e.t_wait[-n:])), 3), "net_enqueue_host_ms": round(float(np.mean(e2e.t_net[-n:])), 3),
                  "stage_ms": {k: round(float(np.mean([s.get(k, 0.0) for s in st])), 3) for k in keys}}))
e2e.close()
