"""Per-phase kernel timeline of the three network nodes (run under `rocprofv3 --kernel-trace`): every phase is bracketed by a marker kernel (torch.erfinv), so the
trace can be cut into  flow | depth | detector trunk | detector heads | label image  and each phase's busy time (sum of kernel durations) compared with its span.
Also prints host-side wall times per phase (synchronised), without needing the profiler."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import synth, pipeline

W, H = 640, 480
ctx = V.Context(device=0, width=W, height=H, max_batch=1)
nodes = pipeline.NetNodes(ctx, H, W)
scene = synth.convoy_scene(4, w=W, h=H, seed=5)
fr = [torch.as_tensor(synth.gray_to_bgr(scene.frame(k)[0]), device="cuda") for k in range(3)]
mk = torch.rand(64, device="cuda") * 0.5
def marker():
    torch.cuda.synchronize(); torch.erfinv(mk); torch.cuda.synchronize()
net = nodes.mask_net
wall = {}
def phase(name, fn, reps=3):
    for _ in range(reps):
        marker(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); wall.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
    return r
for rep in range(2):            # the first pass warms up, the second is the one to read
    wall.clear()
    phase("flow", lambda: nodes.g_flow(fr[0], fr[1]))
    phase("depth", lambda: nodes.g_depth(fr[1]))
    tr = phase("trunk", lambda: nodes.g_trunk(fr[1]))
    feats, logits, deltas = tr
    def heads_rpn():
        return net.rpn.proposals(feats, logits, deltas, (nodes.mask_feed[1], nodes.mask_feed[0]))
    prop, obj = phase("rpn_proposals", heads_rpn)
    def box():
        return net.roi_heads.box(feats[:4], prop, (nodes.mask_feed[1], nodes.mask_feed[0]), obj)
    boxes, scores, labels = phase("box_head", box)
    masks = phase("mask_head", lambda: net.roi_heads.mask(feats[:4], boxes, labels))
    phase("analyse_image_total", lambda: V.nets.analyse_image(net, fr[1], feed=nodes.mask_feed, confidence=nodes.confidence, trunk=nodes.g_trunk))
    marker()
print(json.dumps({k: [round(x, 3) for x in v] for k, v in wall.items()}))
print("detections", int(boxes.shape[0]), "proposals", int((obj >= 0).sum()))
