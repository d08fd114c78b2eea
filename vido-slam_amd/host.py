"""Host-side Python mirror of the reference's operator interface over the C-ABI (include/vido_c.h).

The reference's host is C++ (`VIDO_SLAM::ORBextractor`, `Frame`, `Optimizer` — see the C++ facade under
include/vido_slam/); this module is the ctypes binding the parity tests, bench.py and smoke() drive.
Names follow the reference: `ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)` is
called with a gray image and returns (keypoints, descriptors) like ORBextractor::operator()
(vido_slam/include/ORBextractor.h:39-49).

There is no CPU path here: if libvido_slam_hip.so is missing or no gfx950 device is visible every call
raises VidoError.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIDO_LIB_VARIANT") or os.path.join(_HERE, "libvido_slam_hip.so")      # (VIDO_LIB_VARIANT: an experiment build of the same sources, tools/r6/)

VIDO_OK = 0
KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])


class VidoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vido error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32),
                ("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("compute_descriptors", C.c_int32),
                ("host_threads", C.c_int32)]


_lib = None


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VidoError(-2, "%s not built: run `python __graft_entry__.py` (build()) first; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.vido_last_error.restype = C.c_char_p
    lib.vido_last_error.argtypes = [C.c_void_p]
    lib.vido_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.vido_destroy.argtypes = [C.c_void_p]
    lib.vido_stream.restype = C.c_void_p
    lib.vido_stream.argtypes = [C.c_void_p]
    lib.vido_synchronize.argtypes = [C.c_void_p]
    lib.vido_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.vido_orb_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.vido_orb_extract_color.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.vido_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.vido_orb_read_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vido_orb_read_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.vido_orb_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    lib.vido_hamming_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.vido_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    _lib = lib
    return lib


def default_config(**kw):
    cfg = Config()
    load_library().vido_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One vido_ctx: one HIP stream + device arena on one GPU."""

    def __init__(self, **kw):
        self.lib = load_library()
        self.cfg = default_config(**kw)
        h = C.c_void_p()
        rc = self.lib.vido_create(C.byref(self.cfg), C.byref(h))
        if rc != VIDO_OK:
            raise VidoError(rc, self.lib.vido_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vido_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise VidoError(rc, self.lib.vido_last_error(self.h).decode())
        return rc

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.vido_device_name(self.h, buf, 256))
        return buf.value.decode()

    @property
    def stream(self):
        return self.lib.vido_stream(self.h)

    def synchronize(self):
        self._check(self.lib.vido_synchronize(self.h))

    # ---- ORB -----------------------------------------------------------------------------------
    @property
    def max_kp(self):
        return self.cfg.n_features * 2 + 256

    def orb_extract(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        kps = np.zeros(self.max_kp, KP_DTYPE); desc = np.zeros((self.max_kp, 32), np.uint8); n = C.c_int()
        self._check(self.lib.vido_orb_extract(self.h, _ptr(gray), w, w, h, _ptr(kps), self.max_kp, C.byref(n), _ptr(desc)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def orb_extract_color(self, img, rgb_order=False):
        """cvtColor + ORBextractor::operator() (vido_orb_extract_color): img (h,w,3|4) u8 -> (gray (h,w) u8, keypoints, descriptors)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w, cn = img.shape
        gray = np.empty((h, w), np.uint8)
        kps = np.zeros(self.max_kp, KP_DTYPE); desc = np.zeros((self.max_kp, 32), np.uint8); n = C.c_int()
        self._check(self.lib.vido_orb_extract_color(self.h, _ptr(img), cn, int(bool(rgb_order)), 0, 1, C.c_size_t(0), w * cn, w, h, _ptr(gray),
                                                    _ptr(kps), self.max_kp, C.byref(n), _ptr(desc)))
        return gray, kps[:n.value].copy(), desc[:n.value].copy()

    def orb_extract_batch(self, imgs, want_desc=True, reuse=False):
        """imgs: (n,h,w) u8 numpy array (host) or a (device_ptr, n, h, w, frame_stride, row_stride) tuple.
        reuse=True hands back the same output arrays on every call (no 15 MB allocation per batch)."""
        if isinstance(imgs, tuple):
            ptr, n, h, w, fstride, rstride = imgs
            on_dev = 1
        else:
            imgs = np.ascontiguousarray(imgs, np.uint8)
            n, h, w = imgs.shape
            ptr, fstride, rstride, on_dev = imgs.ctypes.data, h * w, w, 0
        key = (n, bool(want_desc))
        if reuse and getattr(self, "_orb_out", {}).get(key) is not None:
            kps, desc, cnt = self._orb_out[key]
        else:
            kps = np.zeros((n, self.max_kp), KP_DTYPE)
            desc = np.zeros((n, self.max_kp, 32), np.uint8) if want_desc else None
            cnt = np.zeros(n, np.int32)
            if reuse:
                self._orb_out = getattr(self, "_orb_out", {}); self._orb_out[key] = (kps, desc, cnt)
        self._check(self.lib.vido_orb_extract_batch(self.h, C.c_void_p(ptr), on_dev, n, fstride, rstride, w, h, _ptr(kps), self.max_kp,
                                                    _ptr(cnt), _ptr(desc) if want_desc else None))
        return kps, desc, cnt

    def orb_level(self, frame, level, blurred=False):
        lw, lh = C.c_int(), C.c_int()
        self._check(self.lib.vido_orb_level_size(self.h, level, C.byref(lw), C.byref(lh)))
        out = np.empty((lh.value, lw.value), np.uint8)
        self._check(self.lib.vido_orb_read_level(self.h, frame, level, int(blurred), _ptr(out)))
        return out

    def orb_candidates(self, frame, level):
        n = self._check(self.lib.vido_orb_read_candidates(self.h, frame, level, None, 0))
        out = np.empty(max(n, 1), np.uint32)
        self._check(self.lib.vido_orb_read_candidates(self.h, frame, level, _ptr(out), n))
        out = out[:n]
        return (out & 0xfff).astype(np.int32), ((out >> 12) & 0xfff).astype(np.int32), (out >> 24).astype(np.int32)

    def orb_timing(self):
        t = np.zeros(8, np.float32)
        self._check(self.lib.vido_orb_last_timing(self.h, _ptr(t)))
        return dict(zip(["pyramid_ms", "fast_ms", "quadtree_ms", "blur_ms", "orient_brief_ms", "wall_ms", "compact_ms", "n_candidates"], t.tolist()))

    # ---- Hamming ---------------------------------------------------------------------------------
    def hamming_match(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        idx = np.empty(len(a), np.int32); dist = np.empty(len(a), np.int32)
        self._check(self.lib.vido_hamming_match(self.h, _ptr(a), len(a), _ptr(b), len(b), _ptr(idx), _ptr(dist), 0))
        return idx, dist

    def hamming_match_device(self, a_ptr, na, b_ptr, nb, idx_ptr, dist_ptr):
        self._check(self.lib.vido_hamming_match(self.h, C.c_void_p(a_ptr), na, C.c_void_p(b_ptr), nb, C.c_void_p(idx_ptr), C.c_void_p(dist_ptr), 1))


class TrackParams(C.Structure):
    """vido_track_params: dataset/depth/camera constants the reference parses from YAML (Tracking.cc:45-171)."""
    _fields_ = [("dataset", C.c_int32), ("depth_map_factor", C.c_float), ("bf", C.c_float), ("kaist_scale", C.c_float),
                ("th_depth_bg", C.c_float), ("th_depth_obj", C.c_float), ("dense_step", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class FrameLists(C.Structure):
    _fields_ = [("max_stat", C.c_int32), ("max_obj", C.c_int32),
                ("n_stat", C.c_void_p), ("stat_idx", C.c_void_p), ("stat_corr", C.c_void_p), ("stat_flow", C.c_void_p), ("stat_depth", C.c_void_p),
                ("n_obj", C.c_void_p), ("obj_keys", C.c_void_p), ("obj_corr", C.c_void_p), ("obj_depth", C.c_void_p),
                ("obj_label", C.c_void_p), ("obj_flow", C.c_void_p)]


def track_params(dataset=0, depth_map_factor=1.0, bf=387.57, kaist_scale=1.0, th_depth_bg=80.0, th_depth_obj=60.0,
                 dense_step=4, fx=500.0, fy=500.0, cx=320.0, cy=240.0):
    return TrackParams(dataset, depth_map_factor, bf, kaist_scale, th_depth_bg, th_depth_obj, dense_step, fx, fy, cx, cy)


class FrontendView(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("kp_pitch", C.c_int32), ("stat_pitch", C.c_int32), ("obj_pitch", C.c_int32),
                ("kps", C.c_void_p), ("desc", C.c_void_p), ("frame_beg", C.c_void_p),
                ("n_stat", C.c_void_p), ("stat_idx", C.c_void_p), ("stat_corr", C.c_void_p), ("stat_flow", C.c_void_p), ("stat_depth", C.c_void_p),
                ("n_obj", C.c_void_p), ("obj_keys", C.c_void_p), ("obj_corr", C.c_void_p), ("obj_depth", C.c_void_p), ("obj_label", C.c_void_p), ("obj_flow", C.c_void_p)]


class FrameFeatures:
    """Device-side Frame::Frame RGB-D ctor stages (Frame.cc:36-241) bound to a Context."""

    def __init__(self, ctx, params):
        self.ctx, self.p = ctx, params
        self.slots = ctx._check(ctx.lib.vido_track_slots(ctx.h))

    def upload(self, slot0, depth, flow, mask):
        """depth (n,h,w) f32 is rescaled IN PLACE like the reference does to the caller's buffer."""
        assert depth.dtype == np.float32 and depth.flags.c_contiguous
        flow = np.ascontiguousarray(flow, np.float32); mask = np.ascontiguousarray(mask, np.int32)
        n = depth.shape[0] if depth.ndim == 3 else 1
        self.ctx._check(self.ctx.lib.vido_frame_upload(self.ctx.h, slot0, n, _ptr(depth), _ptr(flow), _ptr(mask), 0, C.byref(self.p)))

    def features(self, slot0, kps, n_kps, reuse=False):
        """kps: (n, max_kp) KP_DTYPE, n_kps: (n,) -> dict of per-frame lists (reuse=True: same arrays every call)."""
        kps = np.ascontiguousarray(kps); n_kps = np.ascontiguousarray(n_kps, np.int32)
        n, max_kp = kps.shape
        max_obj = ((self.ctx.cfg.width + 3) // 4) * ((self.ctx.cfg.height + 3) // 4)
        o = getattr(self, "_out", {}).get((n, max_kp)) if reuse else None
        if o is None:
          o = dict(n_stat=np.zeros(n, np.int32), stat_idx=np.zeros((n, max_kp), np.int32), stat_corr=np.zeros((n, max_kp, 2), np.float32),
                 stat_flow=np.zeros((n, max_kp, 2), np.float32), stat_depth=np.zeros((n, max_kp), np.float32),
                 n_obj=np.zeros(n, np.int32), obj_keys=np.zeros((n, max_obj, 2), np.float32), obj_corr=np.zeros((n, max_obj, 2), np.float32),
                 obj_depth=np.zeros((n, max_obj), np.float32), obj_label=np.zeros((n, max_obj), np.int32), obj_flow=np.zeros((n, max_obj, 2), np.float32))
          if reuse:
            self._out = getattr(self, "_out", {}); self._out[(n, max_kp)] = o
        L = FrameLists(max_kp, max_obj, *[o[k].ctypes.data for k in ("n_stat", "stat_idx", "stat_corr", "stat_flow", "stat_depth",
                                                                      "n_obj", "obj_keys", "obj_corr", "obj_depth", "obj_label", "obj_flow")])
        self.ctx._check(self.ctx.lib.vido_frame_features(self.ctx.h, slot0, n, _ptr(kps), _ptr(n_kps), max_kp, C.byref(self.p), C.byref(L)))
        return o

    def frontend_batch(self, slot0, imgs, depth, flow, mask, alias=False):
        """Fused ORB + depth pre-scale + Frame::Frame lists for a batch (vido_frontend_batch).  Every argument is either a
        host numpy array or a (device_ptr, ...) description: imgs like Context.orb_extract_batch, depth/flow/mask either all
        numpy (n,h,w[,2]) or all raw device pointers (ints).  Returns zero-copy numpy VIEWS of the ctx's pinned result
        buffers — valid until the next call on this context.  alias=True (device maps only): no copies into the ctx, the caller keeps the
        three device buffers alive until the slots are overwritten (the reference's own ownership rule)."""
        ctx = self.ctx
        if isinstance(imgs, tuple):
            ptr, n, h, w, fstride, rstride = imgs; img_dev = 1
        else:
            imgs = np.ascontiguousarray(imgs, np.uint8); n, h, w = imgs.shape
            ptr, fstride, rstride, img_dev = imgs.ctypes.data, h * w, w, 0
        if isinstance(depth, np.ndarray):
            assert depth.dtype == np.float32 and depth.flags.c_contiguous
            flow = np.ascontiguousarray(flow, np.float32); mask = np.ascontiguousarray(mask, np.int32)
            dp, fp, mp, maps_dev = depth.ctypes.data, flow.ctypes.data, mask.ctypes.data, 0
        else:
            dp, fp, mp, maps_dev = int(depth), int(flow), int(mask), (2 if alias else 1)     # 2: zero-copy, the slots refer to the caller's device buffers
        v = FrontendView()
        ctx._check(ctx.lib.vido_frontend_batch(ctx.h, C.c_void_p(ptr), img_dev, n, C.c_size_t(fstride), rstride, w, h, C.c_void_p(dp), C.c_void_p(fp), C.c_void_p(mp),
                                               maps_dev, slot0, C.byref(self.p), C.byref(v)))
        cache = self.__dict__.setdefault("_views", {})          # the result buffers are ctx-owned and stable between calls: build each numpy view once
        def arr(p, shape, dtype):
            key = (p, shape, np.dtype(dtype).str)
            a = cache.get(key)
            if a is None:
                count = int(np.prod(shape))
                a = np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p), dtype=dtype, count=count).reshape(shape) if count else np.zeros(shape, dtype)
                if len(cache) > 256: cache.clear()
                cache[key] = a
            return a
        B = ctx.cfg.max_batch if ctx.cfg.max_batch > 1 else 2
        fb = arr(v.frame_beg, (n + 1,), np.int32)
        return dict(kps=arr(v.kps, (n, v.kp_pitch), KP_DTYPE), desc=arr(v.desc, (n, v.kp_pitch, 32), np.uint8), n_kp=np.diff(fb),
                    n_stat=arr(v.n_stat, (n,), np.int32), stat_idx=arr(v.stat_idx, (n, v.stat_pitch), np.int32), stat_corr=arr(v.stat_corr, (n, v.stat_pitch, 2), np.float32),
                    stat_flow=arr(v.stat_flow, (n, v.stat_pitch, 2), np.float32), stat_depth=arr(v.stat_depth, (n, v.stat_pitch), np.float32),
                    n_obj=arr(v.n_obj, (n,), np.int32), obj_keys=arr(v.obj_keys, (n, v.obj_pitch, 2), np.float32), obj_corr=arr(v.obj_corr, (n, v.obj_pitch, 2), np.float32),
                    obj_depth=arr(v.obj_depth, (n, v.obj_pitch), np.float32), obj_label=arr(v.obj_label, (n, v.obj_pitch), np.int32),
                    obj_flow=arr(v.obj_flow, (n, v.obj_pitch, 2), np.float32))

    def gather_static_depth(self, slot, keys):
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2); out = np.empty(len(keys), np.float32)
        self.ctx._check(self.ctx.lib.vido_gather_static_depth(self.ctx.h, slot, _ptr(keys), len(keys), _ptr(out)))
        return out

    def gather_object_depth_label(self, slot, keys):
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2)
        d = np.empty(len(keys), np.float32); l = np.empty(len(keys), np.int32)
        self.ctx._check(self.ctx.lib.vido_gather_object_depth_label(self.ctx.h, slot, _ptr(keys), len(keys), C.c_float(self.p.th_depth_obj), _ptr(d), _ptr(l)))
        return d, l

    def update_mask(self, slot_last, slot_cur, last_label, last_corr):
        last_label = np.ascontiguousarray(last_label, np.int32); last_corr = np.ascontiguousarray(last_corr, np.float32).reshape(-1, 2)
        rec = np.zeros(64, np.int32); nrec = C.c_int32()
        self.ctx._check(self.ctx.lib.vido_update_mask(self.ctx.h, slot_last, slot_cur, _ptr(last_label), _ptr(last_corr), len(last_label),
                                                      _ptr(rec), 64, C.byref(nrec)))
        return rec[:nrec.value].copy()

    def read_maps(self, slot):
        h, w = self.ctx.cfg.height, self.ctx.cfg.width
        d = np.empty((h, w), np.float32); f = np.empty((h, w, 2), np.float32); m = np.empty((h, w), np.int32)
        self.ctx._check(self.ctx.lib.vido_read_maps(self.ctx.h, slot, _ptr(d), _ptr(f), _ptr(m)))
        return d, f, m

    def unproject_world(self, keys, z, Tcw):
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2); z = np.ascontiguousarray(z, np.float32)
        Tcw = np.ascontiguousarray(Tcw, np.float32); out = np.empty((len(z), 3), np.float32)
        self.ctx._check(self.ctx.lib.vido_unproject_world(self.ctx.h, _ptr(keys), _ptr(z), len(z), C.byref(self.p), _ptr(Tcw), _ptr(out)))
        return out

    def scene_flow(self, xyz_last, xyz_cur, sem_last, sem_cur, obj_label):
        a = np.ascontiguousarray(xyz_last, np.float32); b = np.ascontiguousarray(xyz_cur, np.float32)
        sl = np.ascontiguousarray(sem_last, np.int32); sc = np.ascontiguousarray(sem_cur, np.int32)
        ol = np.array(obj_label, np.int32, copy=True); out = np.empty((len(sl), 3), np.float32)
        self.ctx._check(self.ctx.lib.vido_scene_flow(self.ctx.h, _ptr(a), _ptr(b), _ptr(sl), _ptr(sc), len(sl), _ptr(out), _ptr(ol)))
        return out, ol


class PoseProblem(C.Structure):
    """vido_pose_problem (include/vido_c.h)."""
    _fields_ = [("mode", C.c_int32), ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("flow0", C.c_void_p), ("depth", C.c_void_p),
                ("Twl", C.c_double * 16), ("P", C.c_double * 12), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("T_init", C.c_double * 16), ("info_edge", C.c_double), ("info_prior", C.c_double), ("huber_delta", C.c_double),
                ("use_huber", C.c_int32), ("rounds", C.c_int32), ("drop_kernel_after_round", C.c_int32), ("iters", C.c_int32 * 4),
                ("chi2_th", C.c_float * 4)]


class PoseResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("n_inliers", C.c_int32), ("lm_iterations", C.c_int32), ("chi2_final", C.c_double)]


def _fill_pose_problem(p, k, keep):
    n = k["n"]

    def arr(name, cols):
        a = k.get(name)
        if a is None:
            return None
        a = np.ascontiguousarray(a, np.float64).reshape(n, cols) if cols > 1 else np.ascontiguousarray(a, np.float64).reshape(n)
        keep.append(a)
        return a.ctypes.data
    p.mode, p.n = k["mode"], n
    p.Xw, p.obs, p.flow0, p.depth = arr("Xw", 3), arr("obs", 2), arr("flow0", 2), arr("depth", 1)
    p.Twl[:] = list(np.asarray(k.get("Twl", np.eye(4)), np.float64).reshape(16))
    p.P[:] = list(np.asarray(k.get("P", np.zeros((3, 4))), np.float64).reshape(12))
    p.fx, p.fy, p.cx, p.cy = k["fx"], k["fy"], k["cx"], k["cy"]
    p.T_init[:] = list(np.asarray(k["T_init"], np.float64).reshape(16))
    p.info_edge, p.info_prior, p.huber_delta = k["info_edge"], k["info_prior"], k["huber_delta"]
    p.use_huber, p.rounds, p.drop_kernel_after_round = k["use_huber"], k["rounds"], k["drop_kernel_after_round"]
    p.iters[:] = k["iters"]
    p.chi2_th[:] = k["chi2_th"]


class Optimizer:
    """Mirror of the static VIDO_SLAM::Optimizer pose functions (vido_slam/include/Optimizer.h:26-29) over
    vido_pose_optimize*: problems are the dicts built by vido_slam_amd.problems.pose_problem_*."""

    def __init__(self, ctx):
        self.ctx = ctx

    def pose_optimize_batch(self, problems):
        m = len(problems)
        P = (PoseProblem * m)(); R = (PoseResult * m)(); keep = []
        outl = [np.zeros(max(k["n"], 1), np.uint8) for k in problems]
        flows = [np.zeros((max(k["n"], 1), 2), np.float64) for k in problems]
        for i, k in enumerate(problems):
            _fill_pose_problem(P[i], k, keep)
        op = (C.c_void_p * m)(*[o.ctypes.data for o in outl]); fp = (C.c_void_p * m)(*[f.ctypes.data for f in flows])
        self.ctx._check(self.ctx.lib.vido_pose_optimize_batch(self.ctx.h, P, m, R, op, fp))
        return [dict(T=np.array(R[i].T[:]).reshape(4, 4), n_inliers=R[i].n_inliers, lm_iterations=R[i].lm_iterations,
                     chi2_final=R[i].chi2_final, outlier=outl[i][:problems[i]["n"]].astype(bool), flow=flows[i][:problems[i]["n"]])
                for i in range(m)]

    def pose_optimize(self, problem):
        return self.pose_optimize_batch([problem])[0]


class BaProblem(C.Structure):
    """vido_ba_problem (include/vido_c.h)."""
    _fields_ = [("n_cam", C.c_int32), ("n_pt", C.c_int32), ("n_obs", C.c_int32), ("n_odo", C.c_int32), ("prior_cam", C.c_int32),
                ("use_huber", C.c_int32), ("max_iters", C.c_int32), ("pad", C.c_int32),
                ("cam_T", C.c_void_p), ("pt_xyz", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p), ("obs_meas", C.c_void_p),
                ("odo_i", C.c_void_p), ("odo_j", C.c_void_p), ("odo_T", C.c_void_p), ("prior_T", C.c_double * 12),
                ("info_obs", C.c_double), ("info_odo", C.c_double), ("info_prior", C.c_double), ("huber_obs", C.c_double),
                ("huber_odo", C.c_double), ("gain_threshold", C.c_double),
                ("pt_lo", C.c_int32), ("pt_hi", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32)]


class BaResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("lm_trials", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("ms_setup", C.c_double), ("ms_solve_loop", C.c_double), ("ms_linearize_kernel", C.c_double),
                ("ms_schur_kernel", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class _DevBuf:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface, float64)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def torch_allreduce_hook(group=None):
    """All-reduce hook for vido_ba_optimize over torch.distributed (backend nccl == RCCL on ROCm, over xGMI).
    The library hands a device pointer; it is wrapped zero-copy and reduced in place."""
    import torch
    import torch.distributed as dist

    def hook(user, ptr, count, op):
        try:
            t = torch.as_tensor(_DevBuf(ptr, count), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
            torch.cuda.synchronize()
            return 0
        except Exception as e:        # surfaces as VIDO_E_INVALID "all-reduce hook failed"
            import sys
            print("vido all-reduce hook: %r" % (e,), file=sys.stderr)
            return 1
    return ALLREDUCE_FN(hook)


def host_allreduce_hook(group=None):
    """All-reduce hook through HOST memory over a CPU process group (gloo): for `bench.py --oversubscribe` and the one-GPU tests of the N-rank path, where every rank
    shares one device and RCCL cannot be used.  Device -> host copy, gloo all-reduce, host -> device copy."""
    import torch
    import torch.distributed as dist

    def hook(user, ptr, count, op):
        try:
            t = torch.as_tensor(_DevBuf(ptr, count), device="cuda")
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
            t.copy_(h)
            torch.cuda.synchronize()
            return 0
        except Exception as e:
            import sys
            print("vido all-reduce hook (host): %r" % (e,), file=sys.stderr)
            return 1
    return ALLREDUCE_FN(hook)


def rccl_direct_init(ctx, rank=0, world=1):
    """Prepare `ctx` for the library's own RCCL all-reduce (vido_rccl_*, csrc/rccl.cpp): rank 0 draws the unique id, torch.distributed broadcasts its 128 bytes
    (any transport would do), every rank initialises its communicator.  Afterwards pass allreduce="rccl" to ba_optimize."""
    lib = ctx.lib
    lib.vido_rccl_unique_id.argtypes = [C.c_void_p]; lib.vido_rccl_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        ctx._check(lib.vido_rccl_unique_id(buf))
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=0)
        for i, v in enumerate(t.cpu().tolist()):
            buf[i] = v
    ctx._check(lib.vido_rccl_init(ctx.h, buf, rank, world))
    return "rccl"


def landmark_shards(obs_pt, n_pt, world):
    """Contiguous landmark id ranges balanced by observation count."""
    cnt = np.bincount(np.asarray(obs_pt), minlength=n_pt).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(cnt)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * r / world)))
    bounds.append(n_pt)
    bounds = np.maximum.accumulate(np.array(bounds))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]


class BaDynamic(C.Structure):
    _fields_ = [("n_H", C.c_int32), ("n_dyn", C.c_int32), ("n_tern", C.c_int32), ("n_smooth", C.c_int32),
                ("H_T", C.c_void_p), ("dyn_xyz", C.c_void_p), ("dyn_cam", C.c_void_p), ("dyn_meas", C.c_void_p),
                ("tern_prev", C.c_void_p), ("tern_cur", C.c_void_p), ("tern_H", C.c_void_p), ("sm_i", C.c_void_p), ("sm_j", C.c_void_p),
                ("info_dyn", C.c_double), ("info_tern", C.c_double), ("info_smooth", C.c_double),
                ("huber_dyn", C.c_double), ("huber_tern", C.c_double), ("huber_smooth", C.c_double)]


def _ba_dynamic_struct(d):
    b = dict(H_T=np.array(d["H_T"], np.float64).reshape(-1, 12).copy(), dyn_xyz=np.array(d["dyn_xyz"], np.float64).reshape(-1, 3).copy(),
             dyn_cam=np.ascontiguousarray(d["dyn_cam"], np.int32), dyn_meas=np.ascontiguousarray(d["dyn_meas"], np.float64).reshape(-1, 3))
    for name in ("tern_prev", "tern_cur", "tern_H", "sm_i", "sm_j"):
        b[name] = np.ascontiguousarray(d[name], np.int32)
    s = BaDynamic()
    s.n_H, s.n_dyn, s.n_tern, s.n_smooth = len(b["H_T"]), len(b["dyn_cam"]), len(b["tern_prev"]), len(b["sm_i"])
    for name in b:
        setattr(s, name, b[name].ctypes.data)
    for name in ("info_dyn", "info_tern", "info_smooth", "huber_dyn", "huber_tern", "huber_smooth"):
        setattr(s, name, float(d[name]))
    return s, b


def ba_optimize(ctx, k, rank=0, world=1, shard=None, allreduce=None, dynamic=None):
    """Mirror of Optimizer::PartialBatchOptimization / FullBatchOptimization (Optimizer.h:30-31) on the flat problem
    dict built by vido_slam_amd.problems.synth_ba_problem (or by the C++ facade from Map).  Returns the updated
    poses/points and LM statistics.  With world > 1 every rank calls this with its (rank, shard) and the hook."""
    a = dict(cam_T=np.array(k["cam_T"], np.float64).reshape(-1, 12).copy(), pt_xyz=np.array(k["pt_xyz"], np.float64).reshape(-1, 3).copy(),
             obs_cam=np.ascontiguousarray(k["obs_cam"], np.int32), obs_pt=np.ascontiguousarray(k["obs_pt"], np.int32),
             obs_meas=np.ascontiguousarray(k["obs_meas"], np.float64).reshape(-1, 3), odo_i=np.ascontiguousarray(k["odo_i"], np.int32),
             odo_j=np.ascontiguousarray(k["odo_j"], np.int32), odo_T=np.ascontiguousarray(k["odo_T"], np.float64).reshape(-1, 12))
    p = BaProblem()
    p.n_cam, p.n_pt, p.n_obs, p.n_odo = k["n_cam"], k["n_pt"], len(a["obs_cam"]), len(a["odo_i"])
    p.prior_cam, p.use_huber, p.max_iters = k["prior_cam"], k["use_huber"], k["max_iters"]
    for name in ("cam_T", "pt_xyz", "obs_cam", "obs_pt", "obs_meas", "odo_i", "odo_j", "odo_T"):
        setattr(p, name, a[name].ctypes.data)
    p.prior_T[:] = list(np.asarray(k["prior_T"], np.float64).reshape(12))
    for name in ("info_obs", "info_odo", "info_prior", "huber_obs", "huber_odo", "gain_threshold"):
        setattr(p, name, float(k[name]))
    lo, hi = shard if shard is not None else (0, 0)
    p.pt_lo, p.pt_hi, p.rank, p.world = lo, hi, rank, world
    r = BaResult()
    user = None
    if isinstance(allreduce, str) and allreduce == "rccl":            # the library's own RCCL all-reduce on the ctx stream (rccl_direct_init first)
        fn = C.cast(ctx.lib.vido_rccl_allreduce, ALLREDUCE_FN); user = ctx.h
    else:
        fn = allreduce if allreduce is not None else C.cast(None, ALLREDUCE_FN)
    extra = {}
    if dynamic is not None:          # object part of FullBatchOptimization (problems.synth_ba_dynamic)
        s, b = _ba_dynamic_struct(dynamic)
        ctx._check(ctx.lib.vido_ba_optimize_dynamic(ctx.h, C.byref(p), C.byref(s), C.byref(r), fn, user))
        extra = dict(H_T=b["H_T"].reshape(-1, 3, 4), dyn_xyz=b["dyn_xyz"])
    else:
        ctx._check(ctx.lib.vido_ba_optimize(ctx.h, C.byref(p), C.byref(r), fn, user))
    return dict(cam_T=a["cam_T"].reshape(-1, 3, 4), pt_xyz=a["pt_xyz"], iterations=r.iterations, lm_trials=r.lm_trials,
                chi2_initial=r.chi2_initial, chi2_final=r.chi2_final, lambda_final=r.lambda_final, ms_setup=r.ms_setup,
                ms_solve_loop=r.ms_solve_loop, ms_linearize_kernel=r.ms_linearize_kernel, ms_schur_kernel=r.ms_schur_kernel, **extra)


def pnp_ransac(ctx, pts3d, pts2d, K, max_iters=500, reproj_err=0.4, confidence=0.98, seed=1):
    """cv::solvePnPRansac(..., SOLVEPNP_P3P) as called by Tracking::GetInitModelCam/Obj (Tracking.cc:1965, 2068)."""
    X = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); x = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
    T = np.zeros(16); mask = np.zeros(max(len(X), 1), np.uint8); n = C.c_int32()
    ctx._check(ctx.lib.vido_pnp_ransac(ctx.h, _ptr(X), _ptr(x), len(X), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]),
                                       max_iters, C.c_double(reproj_err), C.c_double(confidence), C.c_uint64(seed), _ptr(T), _ptr(mask), C.byref(n)))
    return T.reshape(4, 4), mask[:len(X)].astype(bool), n.value


class NetOps:
    """The reference's native network ops, same names and argument meaning:
    FunctionCorrelation (flow_net/src/correlation/correlation.py:339), layers.ROIAlign / layers.nms
    (maskrcnn_benchmark/layers), BoxCoder.decode (modeling/box_coder.py:52).  numpy in / numpy out (host mode)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def FunctionCorrelation(self, tensorFirst, tensorSecond, intStride):
        a = np.ascontiguousarray(tensorFirst, np.float32); b = np.ascontiguousarray(tensorSecond, np.float32)
        B, Cc, H, W = a.shape
        out = np.empty((B, 49, (H + intStride - 1) // intStride, (W + intStride - 1) // intStride), np.float32)
        self.ctx._check(self.ctx.lib.vido_correlation(self.ctx.h, _ptr(a), _ptr(b), B, Cc, H, W, intStride, _ptr(out), 0))
        return out

    def roi_align(self, input, rois, output_size, spatial_scale, sampling_ratio):
        f = np.ascontiguousarray(input, np.float32); r = np.ascontiguousarray(rois, np.float32).reshape(-1, 5)
        B, Cc, H, W = f.shape; ph, pw = output_size
        out = np.empty((len(r), Cc, ph, pw), np.float32)
        self.ctx._check(self.ctx.lib.vido_roi_align(self.ctx.h, _ptr(f), B, Cc, H, W, _ptr(r), len(r), C.c_float(spatial_scale), ph, pw,
                                                    sampling_ratio, _ptr(out), 0))
        return out

    def nms(self, boxes, scores, nms_thresh):
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4); s = np.ascontiguousarray(scores, np.float32)
        keep = np.empty(max(len(b), 1), np.int32); n = C.c_int32()
        self.ctx._check(self.ctx.lib.vido_nms(self.ctx.h, _ptr(b), _ptr(s), len(b), C.c_float(nms_thresh), _ptr(keep), C.byref(n), 0))
        return keep[:n.value].astype(np.int64)

    def box_decode(self, rel_codes, boxes, weights=(1.0, 1.0, 1.0, 1.0)):
        d = np.ascontiguousarray(rel_codes, np.float32); b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
        w = (C.c_float * 4)(*weights); out = np.empty_like(d)
        self.ctx._check(self.ctx.lib.vido_box_decode(self.ctx.h, _ptr(d), _ptr(b), len(b), d.shape[1] // 4, w, _ptr(out), 0))
        return out


class ORBextractor:
    """Mirror of VIDO_SLAM::ORBextractor (vido_slam/include/ORBextractor.h:39-49): construct with the five
    ctor arguments, call with a CV_8UC1 image, get keypoints + 32-byte descriptors."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width=640, height=480, device=0, max_batch=1):
        self.ctx = Context(device=device, width=width, height=height, max_batch=max_batch, n_features=nfeatures,
                           scale_factor=scaleFactor, n_levels=nlevels, ini_th_fast=iniThFAST, min_th_fast=minThFAST)

    def __call__(self, image, mask=None):
        return self.ctx.orb_extract(image)


# ---- host-side bookkeeping stages of the tracker on flat arrays (include/vido_c.h "Host-side bookkeeping stages"; csrc/trackhost.cpp).  No device work: these
# run without a GPU (the C++ facade's Tracking::RenewFrameInfo / DynObjTracking / Frame::UndistortKeyPoints call the same functions).
class HostMaps(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("depth", C.c_void_p), ("flow", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


def _host_maps(mask, depth, flow):
    mask = np.ascontiguousarray(mask, np.int32); depth = np.ascontiguousarray(depth, np.float32); flow = np.ascontiguousarray(flow, np.float32)
    h, w = mask.shape
    return HostMaps(mask.ctypes.data, depth.ctypes.data, flow.ctypes.data, w, h), (mask, depth, flow)


def _rc(rc, what):
    if rc < 0:
        raise VidoError(rc, what)


def undistort_points(xy, K, dist):
    """Frame::UndistortKeyPoints (Frame.cc:603-633): xy (n,2) f32, K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); out = np.empty_like(xy)
    Kf = np.ascontiguousarray(K, np.float32); d = np.zeros(5, np.float32); d[:len(dist)] = dist
    _rc(load_library().vido_undistort_points(_ptr(xy), len(xy), _ptr(Kf), _ptr(d), _ptr(out)), "undistort_points")
    return out


def renew_static(mask, depth, flow, stat_xy, TM_sta, sample_xy, max_num):
    """Tracking::RenewFrameInfo, static part (Tracking.cc:2973-3075) -> (src index, inlier id, flow) per kept feature."""
    m, keep = _host_maps(mask, depth, flow)
    stat_xy = np.ascontiguousarray(stat_xy, np.float32).reshape(-1, 2); TM = np.ascontiguousarray(TM_sta, np.int32); sample_xy = np.ascontiguousarray(sample_xy, np.float32).reshape(-1, 2)
    cap = len(TM) + len(sample_xy) + 8
    src = np.zeros(cap, np.int32); inl = np.zeros(cap, np.int32); fl = np.zeros((cap, 2), np.float32); n = C.c_int32()
    _rc(load_library().vido_renew_static(C.byref(m), _ptr(stat_xy), len(stat_xy), _ptr(TM), len(TM), _ptr(sample_xy), len(sample_xy), int(max_num), _ptr(src), _ptr(inl), _ptr(fl), cap,
                                         C.byref(n)), "renew_static")
    return src[:n.value], inl[:n.value], fl[:n.value]


def renew_objects(mask, depth, flow, obj_xy, obj_label, inlier_sets, obj_stat, sem_position, mod_label, tmp_xy, tmp_depth, tmp_sem, tmp_flow, tmp_corr, max_num_obj):
    """Tracking::RenewFrameInfo, object part (Tracking.cc:3116-3270)."""
    m, keep = _host_maps(mask, depth, flow)
    obj_xy = np.ascontiguousarray(obj_xy, np.float32).reshape(-1, 2); obj_label = np.ascontiguousarray(obj_label, np.int32)
    off = np.zeros(len(inlier_sets) + 1, np.int32)
    if len(inlier_sets):
        off[1:] = np.cumsum([len(s_) for s_ in inlier_sets])
    ids = np.ascontiguousarray(np.concatenate([np.asarray(s_, np.int32) for s_ in inlier_sets]) if off[-1] else np.zeros(0, np.int32), np.int32)
    st = np.ascontiguousarray(obj_stat, np.uint8); sp = np.ascontiguousarray(sem_position, np.int32); ml = np.ascontiguousarray(mod_label, np.int32)
    txy = np.ascontiguousarray(tmp_xy, np.float32).reshape(-1, 2); td = np.ascontiguousarray(tmp_depth, np.float32); ts = np.ascontiguousarray(tmp_sem, np.int32)
    tf = np.ascontiguousarray(tmp_flow, np.float32).reshape(-1, 2); tc = np.ascontiguousarray(tmp_corr, np.float32).reshape(-1, 2)
    cap = len(ids) + (len(st) + 1) * len(ts) + 8
    o = dict(keys=np.zeros((cap, 2), np.float32), depth=np.zeros(cap, np.float32), sem=np.zeros(cap, np.int32), flow=np.zeros((cap, 2), np.float32), corr=np.zeros((cap, 2), np.float32),
             inlier=np.zeros(cap, np.int32), label=np.zeros(cap, np.int32)); n = C.c_int32()
    _rc(load_library().vido_renew_objects(C.byref(m), _ptr(obj_xy), _ptr(obj_label), len(obj_xy), len(st), _ptr(off), _ptr(ids), _ptr(st), _ptr(sp), _ptr(ml), _ptr(txy), _ptr(td), _ptr(ts),
                                          _ptr(tf), _ptr(tc), len(ts), int(max_num_obj), _ptr(o["keys"]), _ptr(o["depth"]), _ptr(o["sem"]), _ptr(o["flow"]), _ptr(o["corr"]),
                                          _ptr(o["inlier"]), _ptr(o["label"]), cap, C.byref(n)), "renew_objects")
    return {k: v[:n.value] for k, v in o.items()}


def dyn_obj_tracking(sem_label, obj_label, obj_xy, obj_depth, flow3d, last_sem_label, last_sem_position, last_obj_stat, last_mod_label, rows, cols, sf_mg, sf_ds, th_depth_obj, f_id, max_id):
    """Tracking::DynObjTracking (Tracking.cc:1670-1912)."""
    sem = np.ascontiguousarray(sem_label, np.int32); lab = np.array(obj_label, np.int32, copy=True); n = len(sem)
    xy = np.ascontiguousarray(obj_xy, np.float32).reshape(-1, 2); dep = np.ascontiguousarray(obj_depth, np.float32); f3 = np.ascontiguousarray(flow3d, np.float32).reshape(-1, 3)
    lsem = np.ascontiguousarray(last_sem_label, np.int32); lsp = np.ascontiguousarray(last_sem_position, np.int32); lst = np.ascontiguousarray(last_obj_stat, np.uint8)
    lml = np.ascontiguousarray(last_mod_label, np.int32); mid = C.c_int32(max_id); k = C.c_int32()
    off = np.zeros(n + 2, np.int32); ids = np.zeros(n + 1, np.int32); ml = np.zeros(n + 1, np.int32); sp = np.zeros(n + 1, np.int32)
    _rc(load_library().vido_dyn_obj_tracking(_ptr(sem), _ptr(lab), _ptr(xy), _ptr(dep), _ptr(f3), _ptr(lsem), n, _ptr(lsp), _ptr(lst), _ptr(lml), len(lsp), int(rows), int(cols),
                                             C.c_float(sf_mg), C.c_float(sf_ds), C.c_float(th_depth_obj), int(f_id), C.byref(mid), _ptr(off), _ptr(ids), _ptr(ml), _ptr(sp), n + 1, C.byref(k)),
        "dyn_obj_tracking")
    k = k.value
    return dict(obj_label=lab, objects=[ids[off[i]:off[i + 1]].copy() for i in range(k)], mod_label=ml[:k].copy(), sem_position=sp[:k].copy(), max_id=mid.value)
