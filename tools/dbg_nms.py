"""GPU debug: segmented NMS on heavily overlapping boxes vs the oracle; e2e per-stage host timing of NetNodes.infer with and without the tracker thread."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import nets, pipeline, synth
from oracle import pyoracle as O
ctx = V.Context(width=640, height=480, max_batch=1)
ops = nets.HipOps(ctx)
rng = np.random.RandomState(1)
for n, K in ((1000, 1000), (300, 1000), (64, 64), (65, 128), (2000, 2048)):
    cen = rng.uniform(20, 300, (40, 2)); c = cen[rng.randint(0, 40, n)] + rng.normal(0, 4, (n, 2)); wh = rng.uniform(30, 60, (n, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    ref = np.sort(O.nms(b, np.arange(n, 0, -1).astype(np.float32), 0.7))
    bb = np.zeros((K, 4), np.float32); bb[:n] = b
    keep, cnt = ops.nms_segments(torch.from_numpy(bb).cuda(), torch.zeros(1, dtype=torch.int32).cuda(), torch.tensor([n], dtype=torch.int32).cuda(), K, 0.7)
    k = keep[0, :int(cnt[0])].cpu().numpy()
    same = len(k) == len(ref) and np.array_equal(k, ref)
    print("nms n=%d K=%d: kept %d ref %d %s" % (n, K, len(k), len(ref), "equal" if same else "DIFF"), flush=True)
    if not same:
        m = min(len(k), len(ref)); d = np.nonzero(k[:m] != ref[:m])[0]
        print("   first diff at", d[:3], k[max(0, d[0] - 2):d[0] + 3] if len(d) else None, ref[max(0, d[0] - 2):d[0] + 3] if len(d) else None, flush=True)

# ---- e2e timing
nodes = pipeline.NetNodes(ctx, 480, 640)
print("graphs", nodes.g_flow is not None, nodes.graph_error, flush=True)
scene = synth.convoy_scene(12)
frames = [synth.gray_to_bgr(scene.frame(k)[0]) for k in range(10)]
dev = nodes.dev
def one(prev, cur):
    t = [time.perf_counter()]
    f, d, m, l, ev = nodes.infer(prev, cur); t.append(time.perf_counter())
    for e in ev: e.synchronize()
    t.append(time.perf_counter())
    return [round((b - a) * 1e3, 2) for a, b in zip(t, t[1:])]
prev = torch.as_tensor(frames[0], device=dev)
for k in range(1, 6):
    cur = torch.as_tensor(frames[k], device=dev); print("infer alone (enqueue, wait) ms", one(prev, cur), flush=True); prev = cur
# fine-grained: where inside analyse_image
import vido_slam_amd.nets.maskrcnn as MR
cur = torch.as_tensor(frames[6], device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feats, logits, deltas = nodes.g_trunk(cur) if nodes.g_trunk else nodes._trunk_fn(cur); torch.cuda.synchronize(); t1 = time.perf_counter()
    prop, obj = nodes.mask_net.rpn.proposals(feats, logits, deltas, (800, 1088)); torch.cuda.synchronize(); t2 = time.perf_counter()
    bx, sc, lb = nodes.mask_net.roi_heads.box(feats[:4], prop, (800, 1088), obj); torch.cuda.synchronize(); t3 = time.perf_counter()
    mk = nodes.mask_net.roi_heads.mask(feats[:4], bx, lb); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("trunk %.2f rpn %.2f box %.2f (n=%d) mask %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(bx), (t4 - t3) * 1e3), flush=True)
# with a second thread hammering a tracker ctx (ORB extraction in a loop), as in the pipeline
ctx2 = V.Context(width=640, height=480, max_batch=1)
stop = [False]
g = scene.frame(0)[0]
def hammer():
    while not stop[0]:
        ctx2.orb_extract(g)
th = threading.Thread(target=hammer); th.start()
for k in range(6, 10):
    cur = torch.as_tensor(frames[k], device=dev); print("infer with a busy tracker thread ms", one(prev, cur), flush=True); prev = cur
stop[0] = True; th.join()
