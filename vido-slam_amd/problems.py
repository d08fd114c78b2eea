"""Problem builders shared by the host API, bench.py and the parity tests: the constants each of the
reference's four per-frame optimisers hard-codes (information, Huber delta, chi2 thresholds, iteration
counts), as plain dicts, plus seeded synthetic instances.

  PoseOptimizationNew      Optimizer.cc:2180-2334   mode 0: Omega=I, Huber sqrt(0.01f), 1 round x100, chi2>0.01f
  PoseOptimizationFlow2Cam Optimizer.cc:2622-2824   mode 1: Omega=0.1 I, prior 0.3 I, Huber sqrt(0.04f), 4 rounds x100,
                                                    chi2 {0.04f,5.991,5.991,5.991}, kernel dropped after round 2
  PoseOptimizationObjMot   Optimizer.cc:2826-3035   mode 2: Omega=I, no kernel, 1 round x200, chi2>0.01f
  PoseOptimizationFlow2    Optimizer.cc:3037-3253   mode 1: Omega=0.1 I, prior 0.5 I, Huber sqrt(0.04f), 1 round x200
"""
import numpy as np

F32 = np.float32


def _base(mode, n, fx, fy, cx, cy, T_init):
    return dict(mode=mode, n=int(n), fx=float(fx), fy=float(fy), cx=float(cx), cy=float(cy), T_init=np.asarray(T_init, np.float64),
                Xw=None, obs=None, flow0=None, depth=None, Twl=np.eye(4), P=np.zeros((3, 4)))


def pose_problem_new(Xw, obs, K, T_init):
    d = _base(0, len(obs), K[0], K[1], K[2], K[3], T_init)
    d.update(Xw=np.asarray(Xw, np.float64), obs=np.asarray(obs, np.float64), info_edge=1.0, info_prior=0.0,
             huber_delta=float(np.sqrt(F32(0.01))), use_huber=1, rounds=1, drop_kernel_after_round=2, iters=[100, 10, 10, 10],
             chi2_th=[0.01, 5.991, 5.991, 5.991])
    return d


def pose_problem_flow2cam(obs_last, flow, depth, Twl, K, T_init):
    d = _base(1, len(obs_last), K[0], K[1], K[2], K[3], T_init)
    d.update(obs=np.asarray(obs_last, np.float64), flow0=np.asarray(flow, np.float64), depth=np.asarray(depth, np.float64), Twl=np.asarray(Twl, np.float64),
             info_edge=0.1, info_prior=0.3, huber_delta=float(np.sqrt(F32(0.04))), use_huber=1, rounds=4, drop_kernel_after_round=2,
             iters=[100, 100, 100, 100], chi2_th=[0.04, 5.991, 5.991, 5.991])
    return d


def pose_problem_objmot(Xw, obs, K, Tcw, H_init):
    d = _base(2, len(obs), K[0], K[1], K[2], K[3], H_init)
    KK = np.array([[K[0], 0, K[2], 0], [0, K[1], K[3], 0], [0, 0, 1, 0]], np.float64)
    d.update(Xw=np.asarray(Xw, np.float64), obs=np.asarray(obs, np.float64), P=KK @ np.asarray(Tcw, np.float64), info_edge=1.0, info_prior=0.0,
             huber_delta=0.0, use_huber=0, rounds=1, drop_kernel_after_round=2, iters=[200, 100, 100, 100], chi2_th=[0.01, 5.991, 5.991, 5.991])
    return d


def pose_problem_flow2(obs_last, flow, depth, Twl, K, T_init):
    d = pose_problem_flow2cam(obs_last, flow, depth, Twl, K, T_init)
    d.update(info_prior=0.5, rounds=1, iters=[200, 100, 100, 100])
    return d


# ---- seeded synthetic instances ---------------------------------------------------------------------
def se3_exp(u):
    """g2o SE3Quat::exp (omega, upsilon) -> 4x4 (numpy, double); used only to build synthetic ground truth."""
    w = np.asarray(u[:3], np.float64); v = np.asarray(u[3:], np.float64)
    th = np.linalg.norm(w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-5:
        R = np.eye(3) + O + O @ O; V = R
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th**2 * O @ O
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * O + (th - np.sin(th)) / th**3 * O @ O
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = V @ v
    return T


def synth_pose_scene(n, seed=0, K=(520.0, 515.0, 320.0, 240.0), noise_px=0.05, outlier_frac=0.05, motion=(0.01, -0.02, 0.015, 0.05, -0.02, 0.3)):
    """Static points seen from camera 'last' (identity... offset) and 'cur' = exp(motion) * last.
    Returns dict with everything the four problem builders need."""
    rng = np.random.RandomState(seed)
    fx, fy, cx, cy = K
    T_last = se3_exp([0.02, 0.01, -0.01, 0.1, 0.05, -0.2])          # world -> last camera
    T_cur = se3_exp(motion) @ T_last                                # world -> current camera
    uv_last = np.stack([rng.uniform(20, 620, n), rng.uniform(20, 460, n)], 1)
    z = rng.uniform(3.0, 40.0, n)
    Xc = np.stack([(uv_last[:, 0] - cx) * z / fx, (uv_last[:, 1] - cy) * z / fy, z], 1)
    Twl = np.linalg.inv(T_last)
    Xw = Xc @ Twl[:3, :3].T + Twl[:3, 3]
    Xcur = Xw @ T_cur[:3, :3].T + T_cur[:3, 3]
    uv_cur = np.stack([Xcur[:, 0] / Xcur[:, 2] * fx + cx, Xcur[:, 1] / Xcur[:, 2] * fy + cy], 1)
    uv_cur_noisy = uv_cur + rng.normal(0, noise_px, uv_cur.shape)
    nout = int(outlier_frac * n)
    if nout:
        idx = rng.choice(n, nout, replace=False)
        uv_cur_noisy[idx] += rng.uniform(-15, 15, (nout, 2))
    T_init = se3_exp(rng.normal(0, 0.01, 6)) @ T_cur
    return dict(K=K, T_last=T_last, T_cur=T_cur, Twl=Twl, uv_last=uv_last, depth=z, Xw=Xw, uv_cur=uv_cur_noisy, flow=uv_cur_noisy - uv_last, T_init=T_init)


# ---- bundle adjustment ---------------------------------------------------------------------------------
# Constants of the reference's two batch optimisers (sigma^2 are `const float` there, so information =
# 1/double(float(sigma2))):
#   PartialBatchOptimization (Optimizer.cc:191-196,214): cam 1e-4, 3d_sta 16, Huber 0.01 (deltaHuber* are float),
#       prior I/1e-7 when N == WINDOW_SIZE, <=100 iterations, gain threshold 1e-3
#   FullBatchOptimization    (Optimizer.cc:1333-1338,1355): cam 1e-4, 3d_sta 80, prior I*1e5 on frame 0,
#       <=300 iterations, gain threshold 1e-4
def ba_constants(kind="local"):
    if kind == "local":
        return dict(info_obs=1.0 / float(F32(16)), info_odo=1.0 / float(F32(0.0001)), info_prior=1.0 / 0.0000001,
                    huber_obs=float(F32(0.01)), huber_odo=float(F32(0.01)), use_huber=1, max_iters=100, gain_threshold=1e-3)
    return dict(info_obs=1.0 / float(F32(80)), info_odo=1.0 / float(F32(0.0001)), info_prior=1e5,
                huber_obs=float(F32(0.01)), huber_odo=float(F32(0.01)), use_huber=1, max_iters=300, gain_threshold=1e-4)


def _iso(T):
    return np.ascontiguousarray(np.asarray(T, np.float64)[:3, :4])


def synth_ba_problem(n_cam=20, n_pt=2000, seed=7, kind="local", track_len=None, obs_noise=0.02, pose_noise=0.05, with_prior=True,
                     step=1.0):
    """cameras on a forward path with yaw jitter; landmarks in a box ahead of the path;
    measurement = 3-D point in the camera frame + N(0,(obs_noise*z)^2); odometry = true relative pose o exp(N(0,1e-3));
    initial poses perturbed by exp(N(0,pose_noise)).  track_len=None: every landmark is seen from every camera where
    1 < z < 60 (local BA); track_len=k: each landmark is seen in a contiguous run of ~k frames (global BA)."""
    rng = np.random.RandomState(seed)
    cams = []                        # camera-to-world
    T = np.eye(4)
    for i in range(n_cam):
        cams.append(T.copy())
        yaw = np.deg2rad(rng.uniform(-2, 2))
        d = np.eye(4); d[:3, :3] = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]]); d[2, 3] = step
        T = T @ d
    cams = np.stack(cams)
    length = step * (n_cam - 1)
    if track_len is None:
        pts = np.stack([rng.uniform(-20, 20, n_pt), rng.uniform(-5, 5, n_pt), rng.uniform(5, length + 45, n_pt)], 1)
    else:
        zc = np.sort(rng.uniform(0, length, n_pt))        # landmark ids in creation (= temporal) order, as in a SLAM map
        pts = np.stack([rng.uniform(-15, 15, n_pt), rng.uniform(-4, 4, n_pt), zc + rng.uniform(8, 25, n_pt)], 1)
    obs_cam, obs_pt, obs_meas = [], [], []
    inv = np.linalg.inv(cams)
    for c in range(n_cam):
        Xc = pts @ inv[c][:3, :3].T + inv[c][:3, 3]
        vis = (Xc[:, 2] > 1) & (Xc[:, 2] < 60)
        if track_len is not None:
            first = np.clip((pts[:, 2] - 25) / step - track_len / 2, 0, max(n_cam - track_len, 0)).astype(int)
            vis &= (c >= first) & (c < first + track_len)
        idx = np.nonzero(vis)[0]
        noise = rng.normal(0, 1, (len(idx), 3)) * (obs_noise * Xc[idx, 2:3])
        obs_cam.append(np.full(len(idx), c, np.int32)); obs_pt.append(idx.astype(np.int32)); obs_meas.append(Xc[idx] + noise)
    obs_cam = np.concatenate(obs_cam); obs_pt = np.concatenate(obs_pt); obs_meas = np.concatenate(obs_meas)
    # drop landmarks with < 3 observations (tracklet validity, Optimizer.cc:75,86) and re-index
    cnt = np.bincount(obs_pt, minlength=n_pt)
    keep = cnt >= 3
    remap = -np.ones(n_pt, np.int64); remap[keep] = np.arange(keep.sum())
    sel = keep[obs_pt]
    obs_cam, obs_pt, obs_meas = obs_cam[sel], remap[obs_pt[sel]].astype(np.int32), obs_meas[sel]
    pts = pts[keep]
    odo_i = np.arange(n_cam - 1, dtype=np.int32); odo_j = odo_i + 1
    odo = np.stack([_iso(np.linalg.inv(cams[i]) @ cams[i + 1] @ se3_exp(rng.normal(0, 1e-3, 6))) for i in range(n_cam - 1)]) if n_cam > 1 else np.zeros((0, 3, 4))
    cam0 = np.stack([_iso(cams[i] @ se3_exp(rng.normal(0, pose_noise, 6))) if i > 0 or not with_prior else _iso(cams[i]) for i in range(n_cam)])
    pts0 = pts + rng.normal(0, 0.05, pts.shape)
    d = dict(n_cam=n_cam, n_pt=len(pts), cam_T=cam0, pt_xyz=pts0, obs_cam=obs_cam, obs_pt=obs_pt, obs_meas=obs_meas,
             odo_i=odo_i, odo_j=odo_j, odo_T=odo, prior_cam=0 if with_prior else -1, prior_T=_iso(cams[0]),
             cam_true=np.stack([_iso(c) for c in cams]), pt_true=pts)
    d.update(ba_constants(kind))
    return d


def dyn_constants():
    """Optimizer.cc:1333-1338,1358: sigma2_3d_dyn 80, sigma2_obj 100, sigma2_obj_smo 1e-3, Huber deltas 0.01."""
    return dict(info_dyn=1.0 / float(F32(80)), info_tern=1.0 / float(F32(100)), info_smooth=1.0 / float(F32(0.001)),
                huber_dyn=float(F32(0.01)), huber_tern=float(F32(0.01)), huber_smooth=float(F32(0.01)))


def synth_ba_dynamic(base, n_obj=2, pts_per_obj=30, seed=21, obs_noise=0.02, min_len=3, max_len=None, full_tracks=False):
    """Object part of the FullBatchOptimization graph on top of a static problem `base` (synth_ba_problem, kind="global"):
    rigid objects moving with a constant world-frame motion H (p_{k+1} = H p_k), each point tracked over a contiguous
    run of frames.  Mirrors Optimizer.cc:1560-1745: one dynamic vertex per observation (initialised at the noisy
    back-projection through the INITIAL camera pose), its camera edge, a ternary edge to the previous vertex of the
    tracklet and the (object, frame) motion vertex (initialised to identity), smoothness edges between consecutive motion
    vertices of one object from frame 3 on.  full_tracks: every object lives in every frame and every point is tracked through all of them — SURVEY 8(d) row 4(b):
    5 objects x 100 points over 20 keyframes = 10 000 dynamic point vertices, 9 500 ternary edges, 95 motion vertices.  Returns the dict of the vido_ba_dynamic fields (+ H_true)."""
    rng = np.random.RandomState(seed)
    n_cam = base["n_cam"]; cams = np.stack([np.vstack([c, [0, 0, 0, 1]]) for c in base["cam_true"]])
    cam0 = np.stack([np.vstack([c, [0, 0, 0, 1]]) for c in base["cam_T"]])
    H_idx = {}; H_true = []; H_T = []
    dyn_xyz, dyn_cam, dyn_meas, t_prev, t_cur, t_H, sm_i, sm_j = [], [], [], [], [], [], [], []
    for o in range(n_obj):
        yaw = np.deg2rad(rng.uniform(-3, 3))
        H = np.eye(4); H[:3, :3] = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        H[:3, 3] = [rng.uniform(-0.2, 0.2), 0.0, rng.uniform(0.6, 1.2)]
        f0 = int(rng.randint(0, max(1, n_cam // 3))); f1 = int(min(n_cam - 1, f0 + rng.randint(max(min_len, n_cam // 2), n_cam)))
        if full_tracks:
            f0, f1 = 0, n_cam - 1
        centre = cams[f0] @ np.array([rng.uniform(-4, 4), rng.uniform(-0.5, 0.5), rng.uniform(8, 14), 1.0])
        P = centre[:3] + rng.uniform(-1, 1, (pts_per_obj, 3))
        for f in range(max(f0, 1), f1 + 1):
            H_idx[(o, f)] = len(H_true); H_true.append(_iso(H)); H_T.append(_iso(np.eye(4)))
            if f > 2 and (o, f - 1) in H_idx:
                sm_i.append(H_idx[(o, f - 1)]); sm_j.append(H_idx[(o, f)])
        for q in range(pts_per_obj):
            a = int(rng.randint(f0, max(f0 + 1, f1 - min_len + 1))); b = int(min(f1, a + rng.randint(min_len - 1, f1 - f0 + 1)))
            if full_tracks:
                a, b = f0, f1
            if max_len is not None:
                b = min(b, a + max_len - 1)          # bounded tracklet length (the usual case: dynamic points are re-sampled every few frames)
            p = np.append(P[q], 1.0)
            for f in range(f0, a):
                p = H @ p
            prev = -1
            for f in range(a, b + 1):
                if f > a:
                    p = H @ p
                Xc = (np.linalg.inv(cams[f]) @ p)[:3]
                m = Xc + rng.normal(0, 1, 3) * obs_noise * Xc[2]
                idx = len(dyn_cam)
                dyn_cam.append(f); dyn_meas.append(m); dyn_xyz.append((cam0[f] @ np.append(m, 1.0))[:3])
                if prev >= 0:
                    t_prev.append(prev); t_cur.append(idx); t_H.append(H_idx[(o, f)])
                prev = idx
    i32 = lambda v: np.asarray(v, np.int32)
    d = dict(n_H=len(H_T), n_dyn=len(dyn_cam), n_tern=len(t_prev), n_smooth=len(sm_i),
             H_T=np.ascontiguousarray(np.stack(H_T)) if H_T else np.zeros((0, 3, 4)), dyn_xyz=np.ascontiguousarray(np.stack(dyn_xyz)),
             dyn_cam=i32(dyn_cam), dyn_meas=np.ascontiguousarray(np.stack(dyn_meas)), tern_prev=i32(t_prev), tern_cur=i32(t_cur), tern_H=i32(t_H),
             sm_i=i32(sm_i), sm_j=i32(sm_j), H_true=np.stack(H_true) if H_true else np.zeros((0, 3, 4)))
    d.update(dyn_constants())
    return d
