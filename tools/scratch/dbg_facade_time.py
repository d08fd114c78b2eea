import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vido_slam_amd as V
import build as vbuild
from test_facade_gpu import write_clip
n = 16
scene = V.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
with tempfile.TemporaryDirectory() as tmp:
    cfg = write_clip(tmp, scene, n)
    r = subprocess.run([vbuild.build_driver(), cfg, os.path.join(tmp, "poses.txt"), os.path.join(tmp, "res_")], capture_output=True, text=True, timeout=300, env=dict(os.environ, VIDO_TRACK_TIMING="1"))
print(r.stdout[-3000:]); print(r.stderr[-2000:])
