"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does (tests/test_layout.py greps for that)."""
import ctypes as C, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])

def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvido_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB

class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int)]
KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])

class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int), ("n_levels", C.c_int), ("ini_th", C.c_int), ("min_th", C.c_int),
                ("scale_factor", C.c_float), ("scale", C.c_float * 16), ("inv_scale", C.c_float * 16),
                ("n_per_level", C.c_int * 16), ("umax", C.c_int * 16)]

def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)

def orb_params(n_features=2000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7):
    p = OrbParams()
    lib().vo_orb_params_init(C.byref(p), n_features, C.c_float(scale_factor), n_levels, ini_th, min_th)
    return p

def level_size(p, w, h, l):
    lw, lh = C.c_int(), C.c_int()
    lib().vo_level_size(C.byref(p), w, h, l, C.byref(lw), C.byref(lh))
    return lw.value, lh.value

def bgr2gray(img, rgb_order=False):
    img = np.ascontiguousarray(img, np.uint8); h, w, c = img.shape
    out = np.empty((h, w), np.uint8)
    lib().vo_bgr2gray(_p(img), w * c, w, h, c, int(rgb_order), _p(out), w)
    return out

def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    out = np.empty((dh, dw), np.uint8)
    lib().vo_resize_linear_u8(_p(src), w, w, h, _p(out), dw, dw, dh)
    return out

def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    out = np.empty_like(src)
    lib().vo_gaussian_blur7(_p(src), w, w, h, _p(out), w)
    return out

def fast_atan2(y, x):
    f = lib().vo_fast_atan2; f.restype = C.c_float
    return f(C.c_float(y), C.c_float(x))

def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    out = np.empty((w * h, 3), np.int32)
    n = lib().vo_fast9_16(_p(img), w, w, h, threshold, int(nonmax), _p(out), w * h)
    return out[:n].copy()

def fast_score_map(img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().vo_fast_score_map(_p(img), w, w, h, _p(out), w)
    return out

def level_candidates(p, img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    cap = w * h // 4 + 16
    cx = np.empty(cap, np.float32); cy = np.empty(cap, np.float32); cr = np.empty(cap, np.float32)
    n = lib().vo_level_candidates(C.byref(p), _p(img), w, w, h, _p(cx), _p(cy), _p(cr), cap)
    return cx[:n].copy(), cy[:n].copy(), cr[:n].copy()

def distribute_octree(cx, cy, cr, minX, maxX, minY, maxY, N):
    cx = np.ascontiguousarray(cx, np.float32); cy = np.ascontiguousarray(cy, np.float32); cr = np.ascontiguousarray(cr, np.float32)
    out = np.empty(max(len(cx), 1), np.int32)
    n = lib().vo_distribute_octree(_p(cx), _p(cy), _p(cr), len(cx), minX, maxX, minY, maxY, N, _p(out), len(out))
    return out[:n].copy()

def ic_angle(img, x, y, p):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    f = lib().vo_ic_angle; f.restype = C.c_float
    return f(_p(img), w, int(x), int(y), p.umax)

def brief(blurred, x, y, angle):
    img = np.ascontiguousarray(blurred, np.uint8); h, w = img.shape
    d = np.empty(32, np.uint8)
    lib().vo_brief(_p(img), w, int(x), int(y), C.c_float(angle), _p(d))
    return d

def orb_pyramid(p, gray):
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    offs = (C.c_int * 16)()
    total = lib().vo_orb_pyramid(C.byref(p), _p(gray), w, w, h, None, offs)
    buf = np.empty(total, np.uint8)
    lib().vo_orb_pyramid(C.byref(p), _p(gray), w, w, h, _p(buf), offs)
    levels = []
    for l in range(p.n_levels):
        lw, lh = level_size(p, w, h, l)
        levels.append(buf[offs[l]:offs[l] + lw * lh].reshape(lh, lw))
    return levels

def orb_extract(p, gray, cap=None):
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    cap = cap or (p.n_features * 2 + 64)
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); ncand = (C.c_int * 16)()
    n = lib().vo_orb_extract(C.byref(p), _p(gray), w, w, h, _p(kps), _p(desc), cap, ncand)
    assert n <= cap
    return kps[:n].copy(), desc[:n].copy(), list(ncand)[:p.n_levels]

def hamming_match(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    idx = np.empty(len(a), np.int32); dist = np.empty(len(a), np.int32)
    lib().vo_hamming_match(_p(a), len(a), _p(b), len(b), _p(idx), _p(dist))
    return idx, dist


# ---- tracking front-end (track_oracle.c) ----------------------------------------------------------
def depth_prescale(depth, mode, factor, bf, scale):
    d = np.array(depth, np.float32, copy=True)
    lib().vo_depth_prescale(_p(d), d.size, mode, C.c_float(factor), C.c_float(bf), C.c_float(scale))
    return d

def static_candidates(kps, depth, flow, mask, th_depth):
    kps = np.ascontiguousarray(kps); n = len(kps); h, w = depth.shape
    idx = np.empty(n, np.int32); corr = np.empty((n, 2), np.float32); fl = np.empty((n, 2), np.float32); dd = np.empty(n, np.float32)
    depth = np.ascontiguousarray(depth, np.float32); flow = np.ascontiguousarray(flow, np.float32); mask = np.ascontiguousarray(mask, np.int32)
    m = lib().vo_static_candidates(_p(kps), n, _p(depth), _p(flow), _p(mask), w, h, C.c_float(th_depth), _p(idx), _p(corr), _p(fl), _p(dd))
    return idx[:m].copy(), corr[:m].copy(), fl[:m].copy(), dd[:m].copy()

def dense_object_samples(depth, flow, mask, th_obj, step=4):
    h, w = depth.shape; cap = ((w + step - 1) // step) * ((h + step - 1) // step)
    keys = np.empty((cap, 2), np.float32); corr = np.empty((cap, 2), np.float32); od = np.empty(cap, np.float32)
    lab = np.empty(cap, np.int32); fl = np.empty((cap, 2), np.float32)
    depth = np.ascontiguousarray(depth, np.float32); flow = np.ascontiguousarray(flow, np.float32); mask = np.ascontiguousarray(mask, np.int32)
    m = lib().vo_dense_object_samples(_p(depth), _p(flow), _p(mask), w, h, C.c_float(th_obj), step, _p(keys), _p(corr), _p(od), _p(lab), _p(fl), cap)
    return keys[:m].copy(), corr[:m].copy(), od[:m].copy(), lab[:m].copy(), fl[:m].copy()

def gather_static_depth(keys, depth):
    keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2); h, w = depth.shape; out = np.empty(len(keys), np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    lib().vo_gather_static_depth(_p(keys), len(keys), _p(depth), w, h, _p(out))
    return out

def gather_object_depth_label(keys, depth, mask, th_obj):
    keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2); h, w = depth.shape
    d = np.empty(len(keys), np.float32); l = np.empty(len(keys), np.int32)
    depth = np.ascontiguousarray(depth, np.float32); mask = np.ascontiguousarray(mask, np.int32)
    lib().vo_gather_object_depth_label(_p(keys), len(keys), _p(depth), _p(mask), w, h, C.c_float(th_obj), _p(d), _p(l))
    return d, l

def update_mask(last_label, last_corr, mask_last, flow_last, mask_cur):
    last_label = np.ascontiguousarray(last_label, np.int32); last_corr = np.ascontiguousarray(last_corr, np.float32)
    mask_last = np.ascontiguousarray(mask_last, np.int32); flow_last = np.ascontiguousarray(flow_last, np.float32)
    out = np.array(mask_cur, np.int32, copy=True); h, w = out.shape; rec = np.zeros(64, np.int32)
    n = lib().vo_update_mask(_p(last_label), _p(last_corr), len(last_label), _p(mask_last), _p(flow_last), _p(out), w, h, _p(rec), 64)
    return out, rec[:n].copy()

def unproject_world(keys, z, fx, fy, cx, cy, Tcw):
    keys = np.ascontiguousarray(keys, np.float32).reshape(-1, 2); z = np.ascontiguousarray(z, np.float32)
    Tcw = np.ascontiguousarray(Tcw, np.float32); out = np.empty((len(z), 3), np.float32)
    lib().vo_unproject_world(_p(keys), _p(z), len(z), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _p(Tcw), _p(out))
    return out

def scene_flow(xl, xc, sl, sc, obj_label):
    xl = np.ascontiguousarray(xl, np.float32); xc = np.ascontiguousarray(xc, np.float32)
    sl = np.ascontiguousarray(sl, np.int32); sc = np.ascontiguousarray(sc, np.int32)
    ol = np.array(obj_label, np.int32, copy=True); out = np.empty((len(sl), 3), np.float32)
    lib().vo_scene_flow(_p(xl), _p(xc), _p(sl), _p(sc), len(sl), _p(out), _p(ol))
    return out, ol


# ---- per-frame optimisers (opt_oracle.c) -----------------------------------------------------------
class PoseProblem(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("flow0", C.c_void_p), ("depth", C.c_void_p),
                ("Twl", C.c_double * 16), ("P", C.c_double * 12), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("T_init", C.c_double * 16), ("info_edge", C.c_double), ("info_prior", C.c_double), ("huber_delta", C.c_double),
                ("use_huber", C.c_int32), ("rounds", C.c_int32), ("drop_kernel_after_round", C.c_int32), ("iters", C.c_int32 * 4),
                ("chi2_th", C.c_float * 4)]

class PoseResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("n_inliers", C.c_int32), ("lm_iterations", C.c_int32), ("chi2_final", C.c_double)]

def pose_optimize(prob_kwargs):
    """prob_kwargs: dict as produced by vido_slam_amd.pose_problem(...) (numpy arrays + scalars)."""
    k = prob_kwargs; n = k["n"]; keep = []
    def arr(name, cols):
        a = k.get(name)
        if a is None: return None
        a = np.ascontiguousarray(a, np.float64).reshape(n, cols) if cols > 1 else np.ascontiguousarray(a, np.float64).reshape(n)
        keep.append(a); return a.ctypes.data
    p = PoseProblem()
    p.mode, p.n = k["mode"], n
    p.Xw, p.obs, p.flow0, p.depth = arr("Xw", 3), arr("obs", 2), arr("flow0", 2), arr("depth", 1)
    p.Twl[:] = list(np.asarray(k.get("Twl", np.eye(4)), np.float64).reshape(16)); p.P[:] = list(np.asarray(k.get("P", np.zeros((3, 4))), np.float64).reshape(12))
    p.fx, p.fy, p.cx, p.cy = k["fx"], k["fy"], k["cx"], k["cy"]
    p.T_init[:] = list(np.asarray(k["T_init"], np.float64).reshape(16))
    p.info_edge, p.info_prior, p.huber_delta = k["info_edge"], k["info_prior"], k["huber_delta"]
    p.use_huber, p.rounds, p.drop_kernel_after_round = k["use_huber"], k["rounds"], k["drop_kernel_after_round"]
    p.iters[:] = k["iters"]; p.chi2_th[:] = k["chi2_th"]
    r = PoseResult(); outl = np.zeros(max(n, 1), np.uint8); fl = np.zeros((max(n, 1), 2), np.float64)
    lib().vo_pose_optimize(C.byref(p), C.byref(r), _p(outl), _p(fl))
    return dict(T=np.array(r.T[:]).reshape(4, 4), n_inliers=r.n_inliers, lm_iterations=r.lm_iterations, chi2_final=r.chi2_final,
                outlier=outl[:n].astype(bool), flow=fl[:n])


# ---- bundle adjustment (ba_oracle.c) -----------------------------------------------------------------
class BaProblem(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("n_pt", C.c_int32), ("n_obs", C.c_int32), ("n_odo", C.c_int32), ("prior_cam", C.c_int32),
                ("use_huber", C.c_int32), ("max_iters", C.c_int32), ("pad", C.c_int32),
                ("cam_T", C.c_void_p), ("pt_xyz", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p), ("obs_meas", C.c_void_p),
                ("odo_i", C.c_void_p), ("odo_j", C.c_void_p), ("odo_T", C.c_void_p), ("prior_T", C.c_double * 12),
                ("info_obs", C.c_double), ("info_odo", C.c_double), ("info_prior", C.c_double), ("huber_obs", C.c_double),
                ("huber_odo", C.c_double), ("gain_threshold", C.c_double)]

class BaResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("lm_trials", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double)]

def ba_struct(k, cls=BaProblem):
    """dict from vido_slam_amd.problems.synth_ba_problem -> (struct, keepalive arrays); cam_T / pt_xyz are COPIES that the solver updates."""
    a = dict(cam_T=np.array(k["cam_T"], np.float64).reshape(-1, 12).copy(), pt_xyz=np.array(k["pt_xyz"], np.float64).reshape(-1, 3).copy(),
             obs_cam=np.ascontiguousarray(k["obs_cam"], np.int32), obs_pt=np.ascontiguousarray(k["obs_pt"], np.int32),
             obs_meas=np.ascontiguousarray(k["obs_meas"], np.float64).reshape(-1, 3), odo_i=np.ascontiguousarray(k["odo_i"], np.int32),
             odo_j=np.ascontiguousarray(k["odo_j"], np.int32), odo_T=np.ascontiguousarray(k["odo_T"], np.float64).reshape(-1, 12))
    p = cls()
    p.n_cam, p.n_pt, p.n_obs, p.n_odo = k["n_cam"], k["n_pt"], len(a["obs_cam"]), len(a["odo_i"])
    p.prior_cam, p.use_huber, p.max_iters = k["prior_cam"], k["use_huber"], k["max_iters"]
    for name in ("cam_T", "pt_xyz", "obs_cam", "obs_pt", "obs_meas", "odo_i", "odo_j", "odo_T"):
        setattr(p, name, a[name].ctypes.data)
    p.prior_T[:] = list(np.asarray(k["prior_T"], np.float64).reshape(12))
    for name in ("info_obs", "info_odo", "info_prior", "huber_obs", "huber_odo", "gain_threshold"):
        setattr(p, name, float(k[name]))
    return p, a

def ba_optimize(k):
    p, a = ba_struct(k); r = BaResult()
    lib().vo_ba_optimize(C.byref(p), C.byref(r))
    return dict(cam_T=a["cam_T"].reshape(-1, 3, 4), pt_xyz=a["pt_xyz"], iterations=r.iterations, lm_trials=r.lm_trials,
                chi2_initial=r.chi2_initial, chi2_final=r.chi2_final, lambda_final=r.lambda_final)

def ba_chi2(k):
    p, a = ba_struct(k); f = lib().vo_ba_chi2; f.restype = C.c_double
    return f(C.byref(p))

class BaDynamic(C.Structure):
    _fields_ = [("n_H", C.c_int32), ("n_dyn", C.c_int32), ("n_tern", C.c_int32), ("n_smooth", C.c_int32),
                ("H_T", C.c_void_p), ("dyn_xyz", C.c_void_p), ("dyn_cam", C.c_void_p), ("dyn_meas", C.c_void_p),
                ("tern_prev", C.c_void_p), ("tern_cur", C.c_void_p), ("tern_H", C.c_void_p), ("sm_i", C.c_void_p), ("sm_j", C.c_void_p),
                ("info_dyn", C.c_double), ("info_tern", C.c_double), ("info_smooth", C.c_double),
                ("huber_dyn", C.c_double), ("huber_tern", C.c_double), ("huber_smooth", C.c_double)]

def badyn_struct(d, cls=BaDynamic):
    """dict from vido_slam_amd.problems.synth_ba_dynamic -> (struct, keepalive arrays); H_T / dyn_xyz are COPIES the solver updates."""
    a = dict(H_T=np.array(d["H_T"], np.float64).reshape(-1, 12).copy(), dyn_xyz=np.array(d["dyn_xyz"], np.float64).reshape(-1, 3).copy(),
             dyn_cam=np.ascontiguousarray(d["dyn_cam"], np.int32), dyn_meas=np.ascontiguousarray(d["dyn_meas"], np.float64).reshape(-1, 3))
    for name in ("tern_prev", "tern_cur", "tern_H", "sm_i", "sm_j"):
        a[name] = np.ascontiguousarray(d[name], np.int32)
    s = cls()
    s.n_H, s.n_dyn, s.n_tern, s.n_smooth = len(a["H_T"]), len(a["dyn_cam"]), len(a["tern_prev"]), len(a["sm_i"])
    for name in a:
        setattr(s, name, a[name].ctypes.data)
    for name in ("info_dyn", "info_tern", "info_smooth", "huber_dyn", "huber_tern", "huber_smooth"):
        setattr(s, name, float(d[name]))
    return s, a

def badyn_optimize(k, d):
    p, a = ba_struct(k); s, b = badyn_struct(d); r = BaResult()
    lib().vo_badyn_optimize(C.byref(p), C.byref(s), C.byref(r))
    return dict(cam_T=a["cam_T"].reshape(-1, 3, 4), pt_xyz=a["pt_xyz"], H_T=b["H_T"].reshape(-1, 3, 4), dyn_xyz=b["dyn_xyz"], iterations=r.iterations,
                lm_trials=r.lm_trials, chi2_initial=r.chi2_initial, chi2_final=r.chi2_final, lambda_final=r.lambda_final)

def badyn_optimize_sparse(k, d):
    """vo_badyn_optimize_sparse: the same LM loop over the sparse un-eliminated system, the linear solve by scipy's SuperLU (the reference: g2o BlockSolverX + CSparse,
    Optimizer.cc:1318-1324).  For graphs whose dense Hessian does not fit (configs[3](b): 36 690 unknowns)."""
    import scipy.sparse as sp, scipy.sparse.linalg as spl
    p, a = ba_struct(k); s, b = badyn_struct(d); r = BaResult()
    cache = {}
    FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))
    def solve(N, nnz, rows, cols, vals, lam, bp, xp):
        try:
            key = C.addressof(vals.contents)
            rr = np.ctypeslib.as_array(rows, (nnz,)); cc = np.ctypeslib.as_array(cols, (nnz,)); vv = np.ctypeslib.as_array(vals, (nnz,))
            sig = (key, nnz, float(vv[:64].sum()), float(vv[-64:].sum()))
            if cache.get("sig") != sig:                     # one assembly per linearisation, re-used by the lambda trials
                cache["A"] = sp.coo_matrix((vv.copy(), (rr.copy(), cc.copy())), shape=(N, N)).tocsc(); cache["sig"] = sig
            # elimination order = the unknowns reversed (dynamic points, static points, then the poses): the classic bundle-adjustment order, 7x faster here than COLAMD
            perm = np.arange(N - 1, -1, -1)
            A = (cache["A"] + lam * sp.identity(N, format="csc")).tocsc()[perm][:, perm].tocsc()
            x = np.empty(N)
            x[perm] = spl.splu(A, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)).solve(np.ctypeslib.as_array(bp, (N,))[perm].copy())
            if not np.all(np.isfinite(x)):
                return 0
            np.ctypeslib.as_array(xp, (N,))[:] = x
            return 1
        except Exception:                                    # singular factorisation -> the LM loop treats it as a failed trial
            return 0
    cb = FN(solve)
    f = lib().vo_badyn_optimize_sparse; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, FN]
    f(C.byref(p), C.byref(s), C.byref(r), cb)
    return dict(cam_T=a["cam_T"].reshape(-1, 3, 4), pt_xyz=a["pt_xyz"], H_T=b["H_T"].reshape(-1, 3, 4), dyn_xyz=b["dyn_xyz"], iterations=r.iterations,
                lm_trials=r.lm_trials, chi2_initial=r.chi2_initial, chi2_final=r.chi2_final, lambda_final=r.lambda_final)

def badyn_system(k, d):
    """dense (H, b, chi2) of the whole graph; unknown order = cams, Hs, static points, dynamic points."""
    p, a = ba_struct(k); s, b = badyn_struct(d)
    N = 6 * (p.n_cam + s.n_H) + 3 * (p.n_pt + s.n_dyn)
    Hm = np.zeros((N, N)); g = np.zeros(N); f = lib().vo_badyn_system; f.restype = C.c_double
    chi = f(C.byref(p), C.byref(s), _p(Hm), _p(g))
    return Hm, g, chi

def edge_tern(H, pp, pc):
    H = np.ascontiguousarray(H, np.float64).reshape(12); pp = np.ascontiguousarray(pp, np.float64); pc = np.ascontiguousarray(pc, np.float64)
    e = np.zeros(3); Jc = np.zeros((3, 3)); JH = np.zeros((3, 6))
    lib().vo_edge_tern(_p(H), _p(pp), _p(pc), _p(e), _p(Jc), _p(JH))
    return e, Jc, JH

def ba_reduced_system(k, lam, pt_lo=0, pt_hi=None, with_cam_factors=True):
    """(S, r, chi2) of the landmark shard [pt_lo, pt_hi) at the problem's current estimate."""
    p, a = ba_struct(k); n6 = 6 * p.n_cam; pt_hi = p.n_pt if pt_hi is None else pt_hi
    Hcc = np.zeros((n6, n6)); bc = np.zeros(n6); Hpp = np.zeros((p.n_pt, 9)); bp = np.zeros((p.n_pt, 3)); W = np.zeros((p.n_obs + 1, 18))
    chi = C.c_double()
    lib().vo_ba_linearize(C.byref(p), pt_lo, pt_hi, int(with_cam_factors), _p(Hcc), _p(bc), _p(Hpp), _p(bp), _p(W), C.byref(chi))
    S = np.zeros((n6, n6)); r = np.zeros(n6)
    lib().vo_ba_schur(C.byref(p), pt_lo, pt_hi, C.c_double(lam), int(with_cam_factors), _p(Hcc), _p(bc), _p(Hpp), _p(bp), _p(W), _p(S), _p(r))
    return S, r, chi.value

def ba_edge_se3(Z, Xi, Xj):
    Z = np.ascontiguousarray(Z, np.float64); Xj = np.ascontiguousarray(Xj, np.float64)
    e = np.zeros(6); Ji = np.zeros((6, 6)); Jj = np.zeros((6, 6))
    if Xi is None:
        lib().vo_ba_edge_se3(_p(Z), None, _p(Xj), _p(e), None, _p(Jj))
    else:
        Xi = np.ascontiguousarray(Xi, np.float64); lib().vo_ba_edge_se3(_p(Z), _p(Xi), _p(Xj), _p(e), _p(Ji), _p(Jj))
    return e, Ji, Jj

def ba_edge_obs(X, pt, m):
    X = np.ascontiguousarray(X, np.float64); pt = np.ascontiguousarray(pt, np.float64); m = np.ascontiguousarray(m, np.float64)
    e = np.zeros(3); Jc = np.zeros((3, 6)); Jp = np.zeros((3, 3))
    lib().vo_ba_edge_obs(_p(X), _p(pt), _p(m), _p(e), _p(Jc), _p(Jp))
    return e, Jc, Jp

def iso_oplus(X, d):
    X = np.array(X, np.float64).reshape(3, 4).copy(); d = np.ascontiguousarray(d, np.float64)
    lib().vo_iso_oplus(_p(X), _p(d))
    return X


# ---- network ops (nets_oracle.c) ---------------------------------------------------------------------
def correlation(f1, f2, stride):
    f1 = np.ascontiguousarray(f1, np.float32); f2 = np.ascontiguousarray(f2, np.float32); B, Cc, H, W = f1.shape
    out = np.empty((B, 49, (H + stride - 1) // stride, (W + stride - 1) // stride), np.float32)
    lib().vo_correlation(_p(f1), _p(f2), B, Cc, H, W, stride, _p(out))
    return out

def roi_align(feat, rois, scale, PH, PW, sampling):
    feat = np.ascontiguousarray(feat, np.float32); rois = np.ascontiguousarray(rois, np.float32); B, Cc, H, W = feat.shape
    out = np.empty((len(rois), Cc, PH, PW), np.float32)
    lib().vo_roi_align(_p(feat), B, Cc, H, W, _p(rois), len(rois), C.c_float(scale), PH, PW, sampling, _p(out))
    return out

def nms(boxes, scores, thresh):
    boxes = np.ascontiguousarray(boxes, np.float32); scores = np.ascontiguousarray(scores, np.float32)
    keep = np.empty(max(len(boxes), 1), np.int32)
    m = lib().vo_nms(_p(boxes), _p(scores), len(boxes), C.c_float(thresh), _p(keep))
    return keep[:m].copy()

def box_decode(deltas, boxes, weights):
    deltas = np.ascontiguousarray(deltas, np.float32); boxes = np.ascontiguousarray(boxes, np.float32); w = np.ascontiguousarray(weights, np.float32)
    out = np.empty_like(deltas)
    lib().vo_box_decode(_p(deltas), _p(boxes), len(boxes), deltas.shape[1] // 4, _p(w), _p(out))
    return out


# ---- P3P-RANSAC (pnp_oracle.c) -----------------------------------------------------------------------
def pnp_ransac(pts3d, pts2d, K, max_iters=500, thr=0.4, conf=0.98, seed=1, with_ransac_model=False):
    """(T, mask, n): T = the pose cv::solvePnPRansac returns (refit on the inliers), mask / n = the inliers of the winning RANSAC model; with_ransac_model appends that model."""
    X = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); x = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
    T = np.zeros(16); Tr = np.zeros(16); mask = np.zeros(max(len(X), 1), np.uint8)
    f = lib().vo_pnp_ransac_full
    n = f(_p(X), _p(x), len(X), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), max_iters, C.c_double(thr), C.c_double(conf),
          C.c_uint64(seed), _p(T), _p(mask), _p(Tr))
    if with_ransac_model:
        return T.reshape(4, 4), mask[:len(X)].astype(bool), n, Tr.reshape(4, 4)
    return T.reshape(4, 4), mask[:len(X)].astype(bool), n

def p3p(P, j):
    P = np.ascontiguousarray(P, np.float64).reshape(3, 3); j = np.ascontiguousarray(j, np.float64).reshape(3, 3)
    R = np.zeros((4, 9)); t = np.zeros((4, 3))
    n = lib().vo_p3p(_p(P), _p(j), _p(R), _p(t))
    return [(R[k].reshape(3, 3), t[k]) for k in range(n)]


class OracleNetOps:
    """box_decode / nms / roi_align for CPU torch tensors through the C oracle — lets the CPU tests run the torch module
    graphs whose product path only accepts the HIP ops (nets.HipOps)."""

    def __init__(self, o):
        self.o = o

    def box_decode(self, deltas, boxes, weights):
        import numpy as np, torch
        if deltas.shape[0] == 0:
            return deltas.clone()
        return torch.from_numpy(self.o.box_decode(deltas.numpy(), boxes.numpy(), np.asarray(weights, np.float32)))

    def nms(self, boxes, scores, thresh):
        import numpy as np, torch
        if boxes.shape[0] == 0:
            return torch.zeros((0,), dtype=torch.int64)
        return torch.from_numpy(self.o.nms(boxes.numpy(), scores.numpy(), float(thresh)).astype(np.int64))

    def nms_grouped(self, boxes, scores, groups, thresh):
        import numpy as np, torch
        keep = []
        g = groups.numpy()
        for c in np.unique(g):
            idx = np.nonzero(g == c)[0]
            keep.append(idx[self.o.nms(boxes.numpy()[idx], scores.numpy()[idx], float(thresh))])
        return torch.from_numpy(np.sort(np.concatenate(keep)).astype(np.int64)) if keep else torch.zeros((0,), dtype=torch.int64)

    def roi_align(self, feat, rois, output_size, spatial_scale, sampling_ratio):
        import torch
        return torch.from_numpy(self.o.roi_align(feat.numpy(), rois.numpy(), float(spatial_scale), output_size[0], output_size[1], sampling_ratio))


# ---- host-side tracking bookkeeping (trackhost_oracle.c): literal restatements of Tracking::RenewFrameInfo / DynObjTracking / GetStaticTrack / GetDynamicTrackNew,
# Frame::UndistortKeyPoints
class HostMaps(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("depth", C.c_void_p), ("flow", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


def _maps(mask, depth, flow):
    mask = np.ascontiguousarray(mask, np.int32); depth = np.ascontiguousarray(depth, np.float32); flow = np.ascontiguousarray(flow, np.float32)
    h, w = mask.shape
    return HostMaps(mask.ctypes.data, depth.ctypes.data, flow.ctypes.data, w, h), (mask, depth, flow)


def undistort_points(xy, K, dist):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); out = np.empty_like(xy)
    K = np.ascontiguousarray(K, np.float32); d = np.zeros(5, np.float32); d[:len(dist)] = dist
    lib().vo_undistort_points(_p(xy), len(xy), _p(K), _p(d), _p(out))
    return out


def renew_static(mask, depth, flow, stat_xy, TM_sta, sample_xy, max_num):
    m, keep = _maps(mask, depth, flow)
    stat_xy = np.ascontiguousarray(stat_xy, np.float32).reshape(-1, 2); TM = np.ascontiguousarray(TM_sta, np.int32); sample_xy = np.ascontiguousarray(sample_xy, np.float32).reshape(-1, 2)
    cap = len(TM) + len(sample_xy) + 8
    src = np.zeros(cap, np.int32); inl = np.zeros(cap, np.int32); fl = np.zeros((cap, 2), np.float32)
    n = lib().vo_renew_static(C.byref(m), _p(stat_xy), _p(TM), len(TM), _p(sample_xy), len(sample_xy), int(max_num), _p(src), _p(inl), _p(fl), cap)
    assert n <= cap
    return src[:n], inl[:n], fl[:n]


def renew_objects(mask, depth, flow, obj_xy, obj_label, inlier_sets, obj_stat, sem_position, mod_label, tmp_xy, tmp_depth, tmp_sem, tmp_flow, tmp_corr, max_num_obj):
    m, keep = _maps(mask, depth, flow)
    obj_xy = np.ascontiguousarray(obj_xy, np.float32).reshape(-1, 2); obj_label = np.ascontiguousarray(obj_label, np.int32)
    off = np.zeros(len(inlier_sets) + 1, np.int32); off[1:] = np.cumsum([len(s) for s in inlier_sets]) if len(inlier_sets) else 0
    ids = np.ascontiguousarray(np.concatenate([np.asarray(s, np.int32) for s in inlier_sets]) if len(inlier_sets) and off[-1] else np.zeros(0, np.int32), np.int32)
    st = np.ascontiguousarray(obj_stat, np.uint8); sp = np.ascontiguousarray(sem_position, np.int32); ml = np.ascontiguousarray(mod_label, np.int32)
    txy = np.ascontiguousarray(tmp_xy, np.float32).reshape(-1, 2); td = np.ascontiguousarray(tmp_depth, np.float32); ts = np.ascontiguousarray(tmp_sem, np.int32)
    tf = np.ascontiguousarray(tmp_flow, np.float32).reshape(-1, 2); tc = np.ascontiguousarray(tmp_corr, np.float32).reshape(-1, 2)
    cap = len(ids) + (len(st) + 1) * len(ts) + 8
    o = dict(keys=np.zeros((cap, 2), np.float32), depth=np.zeros(cap, np.float32), sem=np.zeros(cap, np.int32), flow=np.zeros((cap, 2), np.float32), corr=np.zeros((cap, 2), np.float32),
             inlier=np.zeros(cap, np.int32), label=np.zeros(cap, np.int32))
    n = lib().vo_renew_objects(C.byref(m), _p(obj_xy), _p(obj_label), len(st), _p(off), _p(ids), _p(st), _p(sp), _p(ml), _p(txy), _p(td), _p(ts), _p(tf), _p(tc), len(ts), int(max_num_obj),
                               _p(o["keys"]), _p(o["depth"]), _p(o["sem"]), _p(o["flow"]), _p(o["corr"]), _p(o["inlier"]), _p(o["label"]), cap)
    assert n <= cap
    return {k: v[:n] for k, v in o.items()}


def dyn_obj_tracking(sem_label, obj_label, obj_xy, obj_depth, flow3d, last_sem_label, last_sem_position, last_obj_stat, last_mod_label, rows, cols, sf_mg, sf_ds, th_depth_obj, f_id, max_id):
    sem = np.ascontiguousarray(sem_label, np.int32); lab = np.array(obj_label, np.int32, copy=True); n = len(sem)
    xy = np.ascontiguousarray(obj_xy, np.float32).reshape(-1, 2); dep = np.ascontiguousarray(obj_depth, np.float32); f3 = np.ascontiguousarray(flow3d, np.float32).reshape(-1, 3)
    lsem = np.ascontiguousarray(last_sem_label, np.int32); lsp = np.ascontiguousarray(last_sem_position, np.int32); lst = np.ascontiguousarray(last_obj_stat, np.uint8)
    lml = np.ascontiguousarray(last_mod_label, np.int32); mid = C.c_int32(max_id)
    off = np.zeros(n + 2, np.int32); ids = np.zeros(n + 1, np.int32); ml = np.zeros(n + 1, np.int32); sp = np.zeros(n + 1, np.int32)
    f = lib().vo_dyn_obj_tracking
    k = f(_p(sem), _p(lab), _p(xy), _p(dep), _p(f3), _p(lsem), n, _p(lsp), _p(lst), _p(lml), len(lsp), int(rows), int(cols), C.c_float(sf_mg), C.c_float(sf_ds), C.c_float(th_depth_obj),
          int(f_id), C.byref(mid), _p(off), _p(ids), _p(ml), _p(sp))
    return dict(obj_label=lab, objects=[ids[off[i]:off[i + 1]].copy() for i in range(k)], mod_label=ml[:k].copy(), sem_position=sp[:k].copy(), max_id=mid.value)


def tracklets(rows, labels=None):
    """rows[i][j] = matched feature of frame i for feature j of frame i+1 (or -1).  Returns (list of [(frame, feature)...], obj_ids or None)."""
    n_rows = len(rows); row_n = np.array([len(r) for r in rows], np.int32); row_off = np.zeros(n_rows + 1, np.int32); row_off[1:] = np.cumsum(row_n)
    TM = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int32) for r in rows]) if row_off[-1] else np.zeros(0, np.int32), np.int32)
    lab = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int32) for r in labels]), np.int32) if labels is not None and row_off[-1] else None
    cap_t = int((TM >= 0).sum()) + 1; cap_p = 2 * cap_t + 2
    off = np.zeros(cap_t + 1, np.int32); pairs = np.zeros((cap_p, 2), np.int32); oid = np.zeros(cap_t, np.int32)
    nt = lib().vo_tracklets(n_rows, _p(row_off), _p(row_n), _p(TM), _p(lab) if lab is not None else None, _p(off), _p(pairs), _p(oid) if lab is not None else None, cap_t, cap_p)
    return [[tuple(int(v) for v in pr) for pr in pairs[off[t]:off[t + 1]]] for t in range(nt)], (oid[:nt].copy() if lab is not None else None)
