"""The tracker ALONE on the bench's scene (System.TrackRGBD on the renderer's maps, no networks): per-stage ms and, under `rocprofv3 --kernel-trace --stats`, the uncontended
durations of the tracker's kernels (inside the pipeline they share the GPU with the convolutions, which stretches the single-workgroup ones several times)."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import vido_slam_amd as V
from vido_slam_amd import synth
from vido_slam_amd.system import System
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = synth.convoy_scene(n + 1, w=640, h=480, seed=5)
tmp = tempfile.mkdtemp(prefix="vido_prof_"); cfg = os.path.join(tmp, "settings.yaml"); bench.write_settings(cfg, scene.K, 640, 480)
slam = System(); slam.Init(cfg, System.RGBD)
acc = {}; t_all = []
for k in range(n):
    g, d, f, m = scene.frame(k)
    t0 = time.perf_counter()
    slam.TrackRGBD(synth.gray_to_bgr(g), np.ascontiguousarray(d, np.float32), np.ascontiguousarray(f, np.float32), np.ascontiguousarray(m, np.int32), None, None, k / 30.0, None, 10 ** 6)
    t_all.append((time.perf_counter() - t0) * 1e3)
    if k >= 25:
        for kk, v in slam.stats().items():
            if kk.startswith("ms_"): acc.setdefault(kk, []).append(v)
print(json.dumps({"frames": n, "wall_ms_per_frame_after_window_fill": round(float(np.mean(t_all[25:])), 3), "stage_ms": {k: round(float(np.mean(v)), 3) for k, v in acc.items()}}))
