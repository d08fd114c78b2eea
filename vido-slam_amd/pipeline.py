"""In-process network -> tracker hand-off.

The reference's realtime demo (src/realtime_demo/src/run_vido.cc:57-171) asks three ROS services for the optical flow (TYPE_32FC2), the
depth (MONO16) and the instance mask (MONO8) of every frame: the image goes to each node over the wire, each node uploads it, downloads its
result and sends it back, and TrackRGBD uploads the three maps again.  Here the networks and the tracker share one device: the frame is
uploaded once, the three networks write device tensors, and `vido_frontend_batch` takes those buffers as they are (maps_on_device = 2: the
tracker's map slots alias the caller's tensors, which this object keeps alive for the two frames the tracker may still read them).
Nothing is quantised on the way except what the reference's interfaces quantise themselves (depth to 16 bit, mask to 8 bit), so the
front-end results are identical to the round trip through host arrays (tests/test_pipeline_gpu.py).
"""
import numpy as np
import torch

from . import nets as _nets
from .host import FrameFeatures


def bgr_to_gray(bgr):
    """cvtColor(BGR2GRAY) in 14-bit fixed point on the device: (B*1868 + G*9617 + R*4899 + 8192) >> 14 (Tracking.cc:327-340; same formula as
    the facade's to_gray)."""
    x = bgr.to(torch.int32)
    return ((x[..., 0] * 1868 + x[..., 1] * 9617 + x[..., 2] * 4899 + 8192) >> 14).to(torch.uint8).contiguous()


class NetFrontEnd:
    """RunNet + the Frame construction of the frame that follows (run_vido.cc:138-171, Frame.cc:41-230).

    push(bgr) -> None for the first frame, then a dict with the front-end lists of the frame (FrameFeatures.frontend_batch views: keypoints,
    descriptors, static candidates, dense object samples) plus `labels` (Mask R-CNN class indices) and `slot` (the tracker map slot the
    frame's depth / flow / mask now live in, for vido_gather_* / vido_update_mask)."""

    def __init__(self, ctx, frame_params, flow_net, depth_net, mask_net, mask_feed=(1088, 800), depth_feed=(192, 640), confidence=0.8, keep_raw=False):
        self.ctx = ctx; self.flow_net, self.depth_net, self.mask_net = flow_net, depth_net, mask_net
        self.ff = FrameFeatures(ctx, frame_params)
        self.mask_feed, self.depth_feed, self.confidence = mask_feed, depth_feed, confidence
        self.dev = next(flow_net.parameters()).device
        self.prev = None
        self.slot = 0
        self._hold = {}                    # slot -> tensors the tracker's slot aliases
        self.keep_raw, self.raw = keep_raw, None      # keep_raw: copies of the network outputs of the last frame (the tracker rescales the depth map in place)

    @torch.no_grad()
    def infer(self, prev_bgr, cur_bgr):
        """The three service calls of RunNet on device tensors: flow HxWx2 f32, depth HxW f32 (MONO16 values), mask HxW i32, labels."""
        flow = _nets.analyse_flow(self.flow_net, prev_bgr, cur_bgr)
        mask_u8, labels = _nets.analyse_image(self.mask_net, cur_bgr, feed=self.mask_feed, confidence=self.confidence)
        depth_u16 = _nets.analyse_depth(self.depth_net, cur_bgr, feed=self.depth_feed)
        return flow.contiguous(), depth_u16.to(torch.float32).contiguous(), mask_u8.to(torch.int32).contiguous(), labels

    @torch.no_grad()
    def push(self, bgr):
        cur = torch.as_tensor(np.ascontiguousarray(bgr, np.uint8)).to(self.dev, non_blocking=True)      # the only upload of the frame
        if self.prev is None:
            self.prev = cur
            return None
        # RunNet's queue entry: the CURRENT image with the flow of the pair (previous, current) and the depth / mask of the current image
        flow, depth, mask, labels = self.infer(self.prev, cur)
        gray = bgr_to_gray(cur)
        if self.keep_raw: self.raw = (flow.clone(), depth.clone(), mask.clone())
        h, w = gray.shape
        torch.cuda.current_stream().synchronize()          # the tracker runs on the ctx's own stream
        slot = self.slot
        out = self.ff.frontend_batch(slot, (gray.data_ptr(), 1, h, w, h * w, w), depth.data_ptr(), flow.data_ptr(), mask.data_ptr(), alias=True)
        self._hold[slot] = (gray, depth, flow, mask)        # alive until the slot is overwritten (two frames later)
        self.slot ^= 1
        self.prev = cur
        out = dict(out); out["labels"] = labels; out["slot"] = slot
        return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# The whole realtime chain of the reference (src/realtime_demo/src/run_vido.cc): RunNet (:131-171, three service calls per frame) feeding
# RunVidoSlam (:229-235, System::TrackRGBD), with the two stages overlapped: the networks of frame k+1 run while frame k is tracked.
import queue as _queue
import threading as _threading
import time as _time
import os as _os

# MIOpen user find-db recorded once on an MI355X with `bench.py --miopen-find` (MIOPEN_USER_DB_PATH pointing here): the measured best solver per convolution shape of the
# three nodes at their feed sizes.  With it the default immediate-mode path picks those solvers without searching (the search costs ~3.5 minutes of start-up; the
# heuristic pick without the db is 0.7 ms per frame slower).  Read by MIOpen when its handle is created, i.e. at the first convolution; an explicit setting wins.
# The db is keyed to one MIOpen build (file name gfx950100.HIP.<major>_<minor>_<patch>_...): it is offered only when this process' MIOpen has that version, MIOpen
# writes new entries into the directory it is given, so a read-only installation gets a private copy under the temp directory; VIDO_NO_MIOPEN_DB=1 opts out.
_MIOPEN_DB = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")


def _offer_miopen_db():
    if "MIOPEN_USER_DB_PATH" in _os.environ or _os.environ.get("VIDO_NO_MIOPEN_DB") or not _os.path.isdir(_MIOPEN_DB):
        return None
    try:
        import torch as _t
        v = int(_t.backends.cudnn.version() or 0)                # MIOpen's version on ROCm builds: major * 1e6 + minor * 1e3 + patch
    except Exception:
        return None
    tag = "HIP.%d_%d_%d_" % (v // 1000000, (v // 1000) % 1000, v % 1000)
    files = [f for f in _os.listdir(_MIOPEN_DB) if tag in f]
    if not files:
        return None
    path = _MIOPEN_DB
    if not _os.access(path, _os.W_OK):
        import shutil, tempfile
        path = _os.path.join(tempfile.gettempdir(), "vido_slam_miopen_db_%d" % _os.getuid())
        _os.makedirs(path, exist_ok=True)
        for f in files:
            if not _os.path.exists(_os.path.join(path, f)):
                shutil.copy(_os.path.join(_MIOPEN_DB, f), path)
    _os.environ["MIOPEN_USER_DB_PATH"] = path
    return path


_offer_miopen_db()


class NetNodes:
    """The three network nodes (flow_net / mono_depth2 / mask_rcnn ROS services, run_vido.cc:142-157) resident on one device, fp32 like
    the reference.  infer(prev_bgr, cur_bgr) enqueues the three forwards — the detector on the caller's stream, LiteFlowNet and MonoDepth2 back to back on ONE side stream
    (streams="flow+depth", the default since round 3: with every network a graph and no host synchronisation left in the detector, the two chains fill each other's
    small-kernel phases: 13.8 ms per frame against 16.8 ms with everything on one stream; a stream per network gives 14.4, LiteFlowNet and MonoDepth2 on a stream EACH 19.0;
    round 2 had measured any overlap slower because its detector head still ran eagerly); the tracker of the previous frame overlaps with all of it (EndToEnd) — and returns device tensors in the tracker's input types (run_vido.cc:28-37: depth MONO16 -> CV_32F,
    mask MONO8 -> CV_32SC1, flow 32FC2).  optimize: frozen batch norms folded into the convolutions + fused HIP epilogues
    (nets/fuse.py); graphs: the static-shape parts (all of LiteFlowNet and MonoDepth2 incl. their resize wrappers, Mask R-CNN's backbone +
    FPN + RPN head) are captured into hipGraphs."""

    def __init__(self, ctx, height=480, width=640, optimize=True, graphs=True, streams="flow+depth", miopen_find=False, seed=1,
                 mask_feed=(1088, 800), depth_feed=(192, 640), confidence=0.8, calibrate_scores=True, static_detector=True):
        self.ctx, self.h, self.w = ctx, height, width
        self.mask_feed, self.depth_feed, self.confidence = mask_feed, depth_feed, confidence
        if miopen_find:
            torch.backends.cudnn.benchmark = True                      # MIOpen find: measure the applicable solvers once per layer shape
        dev = torch.device("cuda", ctx.cfg.device)
        self.dev = dev
        # the nodes run concurrently on three streams: ops that use the ctx's device scratch (correlation partials, NMS bit matrix, paste table) must not share
        # it across streams, so the flow node gets its own ctx; the depth node's fused epilogue uses no scratch and shares the mask node's
        from .host import Context as _Context
        self.ctx2 = _Context(device=ctx.cfg.device, width=ctx.cfg.width, height=ctx.cfg.height, max_batch=1)
        self.ops = ops = _nets.HipOps(ctx)                # Mask R-CNN (ROI-Align, NMS, box decode, paste) + MonoDepth2 epilogues
        self.ops_flow = _nets.HipOps(self.ctx2)           # LiteFlowNet (cost volume, epilogues)
        self.flow_net = _nets.fill_deterministic(_nets.LiteFlowNet(self.ops_flow.correlation, epilogue=self.ops_flow.bias_act_, warp=self.ops_flow.backwarp,
                                                                       fused=None if _os.environ.get("VIDO_LFN_NO_FUSED") else self.ops_flow,
                                                                       pair_batch=not _os.environ.get("VIDO_LFN_NO_PAIR_BATCH")), seed).eval().to(dev)
        self.depth_net = _nets.fill_deterministic(_nets.MonoDepth2(), seed + 1).eval().to(dev)
        self.mask_net = _nets.fill_maskrcnn(_nets.MaskRCNN(ops), seed + 2).eval().to(dev)
        # random-init detector: un-saturate the class scores so that the reference's detections_per_img cap binds (see nets/weights.py); a synthetic textured frame
        self.score_scale = 1.0
        if calibrate_scores:
            g = torch.Generator(device="cpu").manual_seed(seed + 3)
            cal = torch.randint(0, 256, (height // 8, width // 8, 3), generator=g, dtype=torch.uint8).repeat_interleave(8, 0).repeat_interleave(8, 1).to(dev)
            with torch.no_grad():
                self.score_scale = _nets.calibrate_detector_scores(self.mask_net, _nets.maskrcnn.image_to_feed(cal, dev, mask_feed, ops=ops))
        self.folded = 0
        if optimize:
            self.folded = _nets.fold_batchnorm(self.depth_net, ops) + _nets.fold_batchnorm(self.mask_net, ops)
        # streams: False = the three networks back to back on the caller's stream; True = one stream each; "depth" = MonoDepth2 alone on a side stream (its ~120 launches of a
        # few microseconds leave most CUs idle: next to the detector's convolutions they cost next to nothing, while three full networks side by side evict each other's L2 sets)
        if isinstance(streams, str):                              # e.g. "depth", "flow", "flow+depth" (those networks share ONE side stream), "flow,depth" (a side stream each)
            side = {}; self.streams = [None, None, None]
            for grp in streams.split(","):                          # a trailing "!" puts the group's stream at high priority
                hi = grp.endswith("!"); grp = grp.rstrip("!")
                st_ = torch.cuda.Stream(device=dev, priority=-1 if hi else 0)
                for name in grp.split("+"):
                    self.streams[{"flow": 0, "depth": 1, "det": 2}[name.strip()]] = st_
        else:
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(3)] if streams else None
        self.g_flow = self.g_depth = self.g_trunk = self.g_det = None
        self.graph_error = None
        self.last_counts = None
        self.det_overflows = 0                                        # frames whose detection count exceeded the static head's slots (redone through the dynamic head)
        ex = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev)
        self._flow_fn = lambda a, b: _nets.analyse_flow(self.flow_net, a, b)
        self._depth_fn = lambda a: _nets.analyse_depth(self.depth_net, a, feed=self.depth_feed, ops=ops).to(torch.float32)
        self._trunk_fn = lambda a: self.mask_net.trunk(_nets.maskrcnn.image_to_feed(a, dev, self.mask_feed, ops=ops))
        # the whole detector with static shapes (nets/maskrcnn.py: heads_static / analyse_image_static): trunk + device-side RPN selection + box head + fixed-slot
        # post-processing + mask head + label image, no host synchronisation -> ONE hipGraph; returns (mask i32 HxW, labels [cap], n_labels, n_det)
        def _det_fn(a):
            feats, logits, deltas = self._trunk_fn(a)
            img, labels, n_lab, n_det = _nets.analyse_image_static(self.mask_net, feats, logits, deltas, (height, width), feed=self.mask_feed, confidence=self.confidence)
            return img.to(torch.int32), labels, n_lab.to(torch.int32), n_det.to(torch.int32)
        self._det_fn = _det_fn
        with torch.no_grad():
            for _ in range(2):                                          # first calls: MIOpen compiles / finds its kernels
                self._flow_fn(ex, ex); self._depth_fn(ex); self._trunk_fn(ex)
            # the mask head's detection-count buckets: every batch size its convolutions will ever see is compiled now, not in the middle of a sequence
            mh = self.mask_net.roi_heads.mask
            feats = self.mask_net.trunk(_nets.maskrcnn.image_to_feed(ex, dev, self.mask_feed, ops=ops))[0][:4]
            for b in mh.buckets:
                mh(feats, torch.tensor([[10.0, 10.0, 200.0, 300.0]], device=dev).repeat(b, 1), torch.ones(b, dtype=torch.int64, device=dev))
            torch.cuda.synchronize()
            if graphs:
                try:
                    self.g_flow = _nets.Graphed(self._flow_fn, [ex, ex])
                    self.g_depth = _nets.Graphed(self._depth_fn, [ex])
                    self.g_trunk = _nets.Graphed(self._trunk_fn, [ex])
                    if not _os.environ.get("VIDO_NO_MASK_GRAPHS"):
                        mh.capture_buckets(self.g_trunk.static_out[0][:4], _nets.Graphed)      # the mask head per detection-count bucket, over the trunk's static feature maps
                except Exception as e:                                  # capture is an optimisation: report, run eagerly
                    self.graph_error = "%s: %s" % (type(e).__name__, e)
                    self.g_flow = self.g_depth = self.g_trunk = None
                    torch.cuda.synchronize()
                if self.g_trunk is not None and static_detector and not _os.environ.get("VIDO_NO_DET_GRAPH"):
                    try:                                                # on its own: a failure here leaves the three graphs above in place (dynamic head after the trunk graph)
                        self._det_fn(ex); torch.cuda.synchronize()
                        self.g_det = _nets.Graphed(self._det_fn, [ex])
                    except Exception as e:
                        self.graph_error = "detector graph: %s: %s" % (type(e).__name__, e)
                        self.g_det = None
                        torch.cuda.synchronize()

    @torch.no_grad()
    def infer(self, prev_bgr, cur_bgr):
        """prev_bgr / cur_bgr: u8 HxWx3 device tensors.  Returns (flow HxWx2 f32, depth HxW f32, mask HxW i32, labels, events): work enqueued on the caller's stream after
        this call sees the complete outputs (the caller's stream waits for the side stream's events); another stream or the host waits for the three events."""
        cur = torch.cuda.current_stream()
        ss = [s or cur for s in self.streams] if self.streams else [cur, cur, cur]
        for s in ss:
            if s is not cur:
                s.wait_stream(cur)
        with torch.cuda.stream(ss[0]):
            flow = (self.g_flow or self._flow_fn)(prev_bgr, cur_bgr)
            e0 = torch.cuda.Event(); e0.record()
        with torch.cuda.stream(ss[1]):
            depth = (self.g_depth or self._depth_fn)(cur_bgr)
            e1 = torch.cuda.Event(); e1.record()
        with torch.cuda.stream(ss[2]):
            if getattr(self, "skip_detector", False):                   # the literal "flow+depth+track+local-BA" chain of BASELINE's metric text (bench.py extra.e2e_without_detector)
                if getattr(self, "_no_det", None) is None:
                    z = torch.zeros((), dtype=torch.int32, device=self.dev)
                    self._no_det = (torch.zeros((self.h, self.w), dtype=torch.int32, device=self.dev), torch.zeros((1,), dtype=torch.int64, device=self.dev), z, z.clone())
                mask, labels, n_lab, n_det = self._no_det
                self.last_counts = (n_lab, n_det)
            elif self.g_det is not None:                                # one graph replay, nothing synchronised: labels [cap] (0 = unused slot), counts stay on the device
                mask, labels, n_lab, n_det = self.g_det(cur_bgr)
                self.last_counts = (n_lab, n_det)
            else:                                                       # dynamic head: its data-dependent tail synchronises the host while the other two networks run
                mask_u8, labels = _nets.analyse_image(self.mask_net, cur_bgr, feed=self.mask_feed, confidence=self.confidence, trunk=self.g_trunk)
                mask = mask_u8.to(torch.int32); self.last_counts = None
            e2 = torch.cuda.Event(); e2.record()
        for s, e in zip(ss, (e0, e1, e2)):                          # the outputs are valid in stream order on the CALLER's stream, whatever queue produced them
            if s is not cur:
                cur.wait_event(e)
        return flow, depth, mask, labels, (e0, e1, e2)

    def check_conv1x1_range(self):
        """The split-fp16 1x1 convolutions (csrc/conv1x1.hip, the default arithmetic) take activations below 65504; a launch that met a larger one raised the context's range
        flag and its outputs are not valid.  Called where a frame's networks are known to be complete; never seen with the detector's weights (activations of a few hundred)."""
        if hasattr(self.ops, "conv1x1_range_flag") and self.ops.conv1x1_range_flag(reset=True):
            raise RuntimeError("conv1x1: an activation left the range of the split-fp16 arithmetic (|x| >= 65504); the frame's detections are not valid — "
                               "run with VIDO_CONV1X1_ARITH=bf16x3 (fp32's range, twice the matrix work)")

    @torch.no_grad()
    def redo_detector_if_overflowed(self, cur_bgr, n_det_host):
        """The static head stores detections_per_img slots; score ties at the reference's kthvalue cut (box_head/inference.py:131-137) can leave more.  The caller reads
        n_det with the frame's hand-over and, in that (rare) case, recomputes the label image through the dynamic head."""
        cap = self.mask_net.config.detections_per_img
        if n_det_host <= cap:
            return None
        self.det_overflows += 1
        mask_u8, labels = _nets.analyse_image(self.mask_net, cur_bgr, feed=self.mask_feed, confidence=self.confidence, trunk=self.g_trunk)
        return mask_u8.to(torch.int32), labels


class EndToEnd:
    """RunNet || RunVidoSlam on one GPU: the caller pushes BGR frames; the networks of frame k+1 are enqueued (three hipGraph replays) on the network stream while a
    worker thread tracks frame k through the System (the C++ facade: ORB, lists, P3P-RANSAC, the four optimisers, scene flow, object tracking, re-seeding, local BA)
    on the tracker's own stream.

    handover = "device" (default): the frame is uploaded ONCE (BGR, 0.9 MB), the three networks write device tensors, those are parked in a device
    ring (three device-to-device copies in stream order, because the next graph replay overwrites the graphs' static outputs) and System.TrackRGBDDevice takes the
    ring's pointers: the tracker's stream waits for the producer's event ON THE DEVICE, no map crosses PCIe in either direction and the host never waits for the
    networks before it starts enqueuing the tracker (the reference's chain being replaced: src/realtime_demo/src/run_vido.cc:57-171, three service calls with the
    image going out and a map coming back each, then TrackRGBD uploading the three maps again).
    handover = "host": round 2's form — one D2H copy per map into pinned buffers, TrackRGBD(host arrays) uploads them again (kept for A/B measurements).

    feed = "nets": the tracker consumes the networks' outputs.  feed = "given": the networks run at full cost and their outputs are parked exactly as above, but the
    tracker is handed caller-supplied depth / flow / mask maps of the frame (uploaded next to the BGR frame) — for synthetic benchmarks, where random-weight networks
    produce maps without any geometry for the tracker to work on.  Frame k is still tracked only after its three forwards have completed (same event)."""

    RING = 4                                  # buffer sets: the tracker keeps the maps of the previous frame in use (mask / flow of frame k-1 during frame k)

    def __init__(self, nodes, system, n_image=10000, feed="nets", handover="device"):
        self.nodes, self.system, self.n_image, self.feed, self.handover = nodes, system, n_image, feed, handover
        # (round 6) the tracker ADOPTS the ring's map buffers instead of copying them into its own slots: the ring holds RING = 4 frames, the tracker reads frames k and k - 1,
        # the networks write at most two frames ahead.  VIDO_TRACK_COPY_MAPS=1 keeps the copies (6.1 MB device-to-device per frame).
        if handover == "device" and hasattr(system, "SetZeroCopyMaps"):
            system.SetZeroCopyMaps(not _os.environ.get("VIDO_TRACK_COPY_MAPS"))
        self._prefetch = handover == "device" and hasattr(system, "PrefetchImageDevice") and not _os.environ.get("VIDO_TRACK_NO_PREFETCH")
        h, w = nodes.h, nodes.w; dev = nodes.dev
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()
        self.host = [dict(bgr=pin((h, w, 3), torch.uint8), flow=pin((h, w, 2), torch.float32), depth=pin((h, w), torch.float32), mask=pin((h, w), torch.int32),
                          counts=pin((2,), torch.int32)) for _ in range(self.RING)]
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.dev = [dict(bgr=mk((h, w, 3), torch.uint8), flow=mk((h, w, 2), torch.float32), depth=mk((h, w), torch.float32), mask=mk((h, w), torch.int32),
                         gflow=mk((h, w, 2), torch.float32), gdepth=mk((h, w), torch.float32), gmask=mk((h, w), torch.int32)) for _ in range(self.RING)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.q = _queue.Queue(maxsize=1)      # the networks run at most one frame ahead of the tracker (+ the one in flight)
        self.poses, self.stats, self.err = [], [], None
        self.t_net, self.t_track, self.t_wait, self.n_det = [], [], [], []
        self.prev = None; self.k = 0
        self.net_lock = _threading.Lock()     # the networks' host-side state (stream adoption, ctx scratch, static trunk buffers) has ONE user at a time: push() or the overflow redo
        self.worker = _threading.Thread(target=self._track_loop, daemon=True); self.worker.start()

    def _track_loop(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            k, slot, ev, img_ev = item
            try:
                if self.err is not None:                                 # a failed frame stops the sequence: later frames are drained, not tracked on a broken System
                    continue
                t0 = _time.perf_counter()
                hb, db = self.host[slot], self.dev[slot]
                T = None
                if self.handover == "device":
                    # (round 6) the ORB extraction needs nothing but the image: it goes onto the tracker's stream BEFORE the wait for the frame's networks and runs beside their
                    # tail, instead of starting — as the frame's first heavy tracker work — against the just-enqueued networks of frame k + 1 (VIDO_TRACK_NO_PREFETCH=1: as before)
                    if self._prefetch and not _os.environ.get("VIDO_E2E_SKIP_TRACK"):
                        self.system.PrefetchImageDevice(db["bgr"].data_ptr(), 3, self.nodes.w, self.nodes.h, img_ev.cuda_event)
                    if self.nodes.g_det is not None:                     # the static head's overflow flag (rare): needs the frame's counts, i.e. a host wait for this one event
                        ev.synchronize()
                        self._redo_if_overflowed(slot)
                        self.nodes.check_conv1x1_range()                 # (the frame's networks are complete: ev)
                    t1 = _time.perf_counter()
                    if not _os.environ.get("VIDO_E2E_SKIP_TRACK"):
                        d, f, m = (db["depth"], db["flow"], db["mask"]) if self.feed == "nets" else (db["gdepth"], db["gflow"], db["gmask"])
                        T = self.system.TrackRGBDDevice(db["bgr"].data_ptr(), 3, self.nodes.w, self.nodes.h, d.data_ptr(), f.data_ptr(), m.data_ptr(), ev.cuda_event,
                                                        float(k), self.n_image)
                else:
                    ev.synchronize()                                     # networks + hand-over copies of frame k are complete
                    self.nodes.check_conv1x1_range()
                    t1 = _time.perf_counter()
                    if self.feed == "nets":
                        d, f, m = hb["depth"].numpy(), hb["flow"].numpy(), hb["mask"].numpy()
                    else:
                        d, f, m = hb["given"]
                    if not _os.environ.get("VIDO_E2E_SKIP_TRACK"):       # diagnosis only: networks + hand-over without the tracker
                        T = self.system.TrackRGBD(hb["bgr"].numpy(), d, f, m, None, None, float(k), None, self.n_image)
                t2 = _time.perf_counter()
                self.poses.append(T); self.stats.append(self.system.stats() if T is not None else {})
                self.t_wait.append((t1 - t0) * 1e3); self.t_track.append((t2 - t1) * 1e3)
                self.n_det.append(int(hb["counts"][1]) if self.nodes.g_det is not None else int(getattr(self.nodes.mask_net.roi_heads.mask, "last_n", 0)))
            except Exception as e:                                       # surfaced by push() / finish()
                self.err = e
            finally:
                self.q.task_done()

    @torch.no_grad()
    def _redo_if_overflowed(self, slot):
        n_det = int(self.host[slot]["counts"][1])
        if n_det <= self.nodes.mask_net.config.detections_per_img:
            return
        # on the NETWORK stream (the detector's HIP ops share one device scratch, which stream order protects) and under the lock push() holds around infer():
        # the host-side state of the nodes (adopted stream, ctx scratch pointers, static trunk buffers) is not protected by stream order
        with self.net_lock, torch.cuda.stream(self.net_stream):
            r = self.nodes.redo_detector_if_overflowed(self.dev[slot]["bgr"], n_det)
            if r is not None:
                self.dev[slot]["mask"].copy_(r[0])
            self.net_stream.synchronize()

    @torch.no_grad()
    def push(self, bgr, given=None):
        """bgr: HxWx3 u8 numpy.  given = (depth f32 HxW raw sensor units, flow f32 HxWx2, mask i32 HxW) for feed == "given" (copied: the caller's arrays are not modified,
        although the tracker rescales the depth map it is handed in place)."""
        if self.err is not None:
            raise self.err
        t0 = _time.perf_counter()
        self.net_stream = torch.cuda.current_stream()
        slot = self.k % self.RING
        hb, db = self.host[slot], self.dev[slot]
        hb["bgr"].numpy()[...] = bgr
        cur = db["bgr"]
        cur.copy_(hb["bgr"], non_blocking=True)                          # the only upload of the frame on the network side
        img_ev = torch.cuda.Event(); img_ev.record()                     # the image is on the device: all the tracker's ORB extraction needs (prefetched in _track_loop)
        if given is not None:
            hb["depth"].numpy()[...] = given[0]; hb["flow"].numpy()[...] = given[1]; hb["mask"].numpy()[...] = given[2]
            if self.handover == "device":                                # the stand-in maps go up next to the frame (feed == "nets" uploads nothing but the frame)
                db["gdepth"].copy_(hb["depth"], non_blocking=True); db["gflow"].copy_(hb["flow"], non_blocking=True); db["gmask"].copy_(hb["mask"], non_blocking=True)
            else:
                hb["given"] = (hb["depth"].numpy().copy(), hb["flow"].numpy().copy(), hb["mask"].numpy().copy())
        prev = cur if self.prev is None else self.prev                   # first frame: RunNet has no previous image yet; the tracker ignores the flow of frame 0's predecessor
        with self.net_lock:
            flow, depth, mask, labels, evs = self.nodes.infer(prev, cur)
        # graph outputs are static buffers that the next replay overwrites: park them in this slot's device buffers (three device-to-device copies of 4.8 MB in stream order)
        if self.nodes.streams is not None:
            for e in evs:
                torch.cuda.current_stream().wait_event(e)
        db["flow"].copy_(flow, non_blocking=True); db["depth"].copy_(depth, non_blocking=True); db["mask"].copy_(mask, non_blocking=True)
        if self.nodes.last_counts is not None:
            hb["counts"][0:1].copy_(self.nodes.last_counts[0].reshape(1), non_blocking=True); hb["counts"][1:2].copy_(self.nodes.last_counts[1].reshape(1), non_blocking=True)
        parked = torch.cuda.Event(); parked.record()
        done = parked
        if self.handover != "device":                                    # round 2's hand-over: the maps go to pinned host buffers on the copy stream
            cs = self.copy_stream
            cs.wait_event(parked)
            with torch.cuda.stream(cs):
                if self.feed == "nets":
                    hb["flow"].copy_(db["flow"], non_blocking=True); hb["depth"].copy_(db["depth"], non_blocking=True); hb["mask"].copy_(db["mask"], non_blocking=True)
                else:                                                    # same traffic, into a scratch set (hb[...] holds the given maps)
                    self._sink = getattr(self, "_sink", None) or [torch.empty_like(hb["flow"]).pin_memory(), torch.empty_like(hb["depth"]).pin_memory(), torch.empty_like(hb["mask"]).pin_memory()]
                    self._sink[0].copy_(db["flow"], non_blocking=True); self._sink[1].copy_(db["depth"], non_blocking=True); self._sink[2].copy_(db["mask"], non_blocking=True)
                done = torch.cuda.Event(); done.record()
        self._alive = (flow, depth, mask, labels)
        self.prev = cur
        self.t_net.append((_time.perf_counter() - t0) * 1e3)
        self.q.put((self.k, slot, done, img_ev))                         # blocks while the tracker is still two frames behind
        self.k += 1

    def finish(self):
        self.q.join()
        if self.err is not None:
            raise self.err

    def close(self):
        self.q.put(None)
        self.worker.join(timeout=10)
