"""The three network nodes as the pipeline runs them (pipeline.NetNodes at 640x480: folded batch norms, fused HIP passes, static detector head), EAGER (no hipGraph), two
frames, for `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace`; tools/nets_pmc.py --summarise turns the counter csv into nets_mfma.json."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd import synth, pipeline
ctx = V.Context(device=0, width=640, height=480, max_batch=1)
nodes = pipeline.NetNodes(ctx, 480, 640, graphs=False)
scene = synth.convoy_scene(4, w=640, h=480, seed=5)
fr = [torch.as_tensor(synth.gray_to_bgr(scene.frame(k)[0]), device="cuda") for k in range(3)]
with torch.no_grad():
    for k in (1, 2):
        nodes._flow_fn(fr[k - 1], fr[k]); nodes._depth_fn(fr[k]); nodes._det_fn(fr[k])
torch.cuda.synchronize()
