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
