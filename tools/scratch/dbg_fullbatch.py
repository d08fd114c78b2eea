import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vido_slam_amd as V
import build as vbuild
from test_facade_gpu import write_clip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
scene = V.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05), (2.5, 0.3, 12.0, -0.2, 0.0, 0.1)))
if len(sys.argv) > 2:          # only write the clip + print the driver command (for rocprofv3)
    os.makedirs(sys.argv[2], exist_ok=True)
    cfg = write_clip(sys.argv[2], scene, n, dataset=2, factor=256.0)
    print(vbuild.build_driver(), cfg, os.path.join(sys.argv[2], "poses.txt"), os.path.join(sys.argv[2], "res_")); sys.exit(0)
with tempfile.TemporaryDirectory() as tmp:
    cfg = write_clip(tmp, scene, n, dataset=2, factor=256.0)
    r = subprocess.run([vbuild.build_driver(), cfg, os.path.join(tmp, "poses.txt"), os.path.join(tmp, "res_")], capture_output=True, text=True, timeout=600, env=dict(os.environ, VIDO_BA_VERBOSE="1"))
    print(r.stdout[-1500:]); print("\n".join(r.stderr.splitlines()[-6:]))
