"""A/B: incremental tracklet store vs the previous full-rebuild build (vido-slam_amd/_old/) on the same clip; all result files must be byte-identical."""
import sys, os, subprocess, tempfile, filecmp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vido_slam_amd as V
import build as vbuild
from test_facade_gpu import write_clip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
old = os.path.join(ROOT, "vido-slam_amd", "_old")
for dataset, factor in ((1, 1.0), (2, 256.0)):
    scene = V.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05), (2.5, 0.3, 12.0, -0.2, 0.0, 0.1)))
    with tempfile.TemporaryDirectory() as tmp:
        cfg = write_clip(tmp, scene, n, dataset=dataset, factor=factor) if dataset != 1 else write_clip(tmp, scene, n)
        outs = {}
        for tag, drv, env in (("new", vbuild.build_driver(), os.environ), ("old", os.path.join(old, "run_vido_slam.bin"), dict(os.environ, LD_LIBRARY_PATH=old + ":" + os.environ.get("LD_LIBRARY_PATH", "")))):
            d = os.path.join(tmp, tag); os.makedirs(d)
            r = subprocess.run([drv, cfg, os.path.join(d, "poses.txt"), os.path.join(d, "res_")], capture_output=True, text=True, timeout=600, env=env)
            print(tag, dataset, r.returncode, "|", " ; ".join(l for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("tracklets", "stage_ms", "track_ms", "frames")), r.stderr[-500:])
            outs[tag] = d
        files = sorted(os.listdir(outs["new"]))
        same = [(f, filecmp.cmp(os.path.join(outs["new"], f), os.path.join(outs["old"], f), shallow=False)) for f in files]
        print("dataset", dataset, same)
        assert all(s for _, s in same), same
print("AB OK")
