"""`.g2o` text interchange for the batch graphs.

The reference dumps the full-batch graph before / after optimisation with g2o's `SparseOptimizer::save`
(Optimizer.cc:1937-1939); the facade writes the same tags (`csrc/facade.cpp: dump_g2o`).  This module reads such a file
back into the flat problem dictionaries `host.ba_optimize` takes, and writes them out again, so a graph produced by
the reference (or by any other g2o front end that uses these vertex / edge types) can be solved here and vice versa.

Tags (the vendored g2o's `types/slam3d`, `types/types_dyn_slam3d.cpp`):
  PARAMS_SE3OFFSET id  x y z qx qy qz qw
  VERTEX_SE3:QUAT  id  x y z qx qy qz qw                     camera-to-world pose | object motion H
  VERTEX_TRACKXYZ  id  x y z                                   static landmark | dynamic point (one per observation)
  EDGE_SE3_PRIOR   id offset  x y z qx qy qz qw  info(21)      prior on the first camera
  EDGE_SE3:QUAT    i j  x y z qx qy qz qw  info(21)            odometry (cameras) | motion smoothness (H, identity measurement)
  EDGE_SE3_TRACKXYZ cam pt offset  x y z  info(6)              3-D point in the camera frame
  EDGE_SE3_MOTION  p_prev p_cur H  x y z  info(6)              LandmarkMotionTernaryEdge, e = p_prev - H^-1 p_cur
Robust kernels are not part of the format: Huber widths are arguments of read_g2o (defaults = FullBatchOptimization's).

Vertex roles are recovered from the edges: an SE(3) vertex that observes a point or carries the prior is a camera, one that
is the third vertex of a motion edge is an object motion, the rest inherit the role of their EDGE_SE3:QUAT neighbours; a
point is dynamic when a motion edge touches it.
"""
import numpy as np


def quat_to_rot(q):
    """(qx, qy, qz, qw) -> 3x3 (Eigen Quaterniond::toRotationMatrix after normalisation)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    """3x3 -> (qx, qy, qz, qw), w >= 0 (Eigen's Quaternion(R) branches + g2o's normalize)."""
    R = np.asarray(R, np.float64); t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0); q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    q /= np.linalg.norm(q)
    return -q if q[3] < 0 else q


def _se3(vals):
    T = np.zeros((3, 4)); T[:, :3] = quat_to_rot(vals[3:7]); T[:, 3] = vals[:3]
    return T.reshape(12)


def _uniform_info(vals, n, what):
    """Upper-triangular information block -> the scalar s of s*I (the only form the flat problem carries)."""
    m = np.zeros((n, n)); m[np.triu_indices(n)] = vals
    d = np.diag(m)
    if not (np.allclose(m - np.diag(d), 0) and np.allclose(d, d[0], rtol=1e-6)):
        raise ValueError("g2o: %s information is not a multiple of the identity" % what)
    return float(d[0])


def read_g2o(path, huber=0.01, max_iters=300, gain_threshold=1e-4, use_huber=1):
    """Returns (problem, dynamic or None, ids).  ids = dict(cam=[file id per camera], pt=[...], H=[...], dyn=[...]) in solver order."""
    se3, xyz = {}, {}
    prior, e_se3, e_trk, e_mot = [], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t: continue
            tag = t[0]; v = t[1:]
            if tag == "VERTEX_SE3:QUAT": se3[int(v[0])] = np.array(v[1:8], np.float64)
            elif tag == "VERTEX_TRACKXYZ": xyz[int(v[0])] = np.array(v[1:4], np.float64)
            elif tag == "EDGE_SE3_PRIOR": prior.append((int(v[0]), np.array(v[2:9], np.float64), np.array(v[9:30], np.float64)))
            elif tag == "EDGE_SE3:QUAT": e_se3.append((int(v[0]), int(v[1]), np.array(v[2:9], np.float64), np.array(v[9:30], np.float64)))
            elif tag == "EDGE_SE3_TRACKXYZ": e_trk.append((int(v[0]), int(v[1]), np.array(v[3:6], np.float64), np.array(v[6:12], np.float64)))
            elif tag == "EDGE_SE3_MOTION": e_mot.append((int(v[0]), int(v[1]), int(v[2]), np.array(v[6:12], np.float64)))
            elif tag in ("PARAMS_SE3OFFSET", "FIX"): continue
            else: raise ValueError("g2o: unsupported tag %s" % tag)
    if len(prior) > 1: raise ValueError("g2o: more than one EDGE_SE3_PRIOR")
    # roles of the SE(3) vertices
    role = {}
    for c, _, _, _ in e_trk: role[c] = "cam"
    for c, _, _ in prior: role[c] = "cam"
    for _, _, h, _ in e_mot:
        if role.get(h) == "cam": raise ValueError("g2o: vertex %d is both a camera and an object motion" % h)
        role[h] = "H"
    changed = True
    while changed:
        changed = False
        for i, j, _, _ in e_se3:
            if i in role and j not in role: role[j] = role[i]; changed = True
            elif j in role and i not in role: role[i] = role[j]; changed = True
    for k in se3: role.setdefault(k, "cam")
    cams = sorted(k for k in se3 if role[k] == "cam"); Hs = sorted(k for k in se3 if role[k] == "H")
    dyn_set = set()
    for a, b, _, _ in e_mot: dyn_set.add(a); dyn_set.add(b)
    pts = sorted(k for k in xyz if k not in dyn_set); dyns = sorted(dyn_set)
    ci = {k: n for n, k in enumerate(cams)}; hi = {k: n for n, k in enumerate(Hs)}
    pi = {k: n for n, k in enumerate(pts)}; di = {k: n for n, k in enumerate(dyns)}
    pr = dict(n_cam=len(cams), n_pt=len(pts), cam_T=np.array([_se3(se3[k]) for k in cams]).reshape(-1, 12),
              pt_xyz=np.array([xyz[k] for k in pts], np.float64).reshape(-1, 3), use_huber=use_huber, max_iters=max_iters, gain_threshold=gain_threshold,
              huber_obs=huber, huber_odo=huber, prior_cam=-1, prior_T=np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64), info_prior=0.0, info_obs=1.0, info_odo=1.0)
    if prior:
        c, m, inf = prior[0]; pr["prior_cam"] = ci[c]; pr["prior_T"] = _se3(m); pr["info_prior"] = _uniform_info(inf, 6, "prior")
    oc, op, om, dc, dm, dyn_of = [], [], [], [], [], {}
    info_obs = info_dyn = None
    for c, p, m, inf in e_trk:
        s = _uniform_info(inf, 3, "point observation")
        if p in di:
            if p in dyn_of: raise ValueError("g2o: dynamic point %d has two observations" % p)
            dyn_of[p] = (ci[c], m); info_dyn = s if info_dyn is None else info_dyn
            if not np.isclose(s, info_dyn, rtol=1e-6): raise ValueError("g2o: dynamic observations with different information")
        else:
            oc.append(ci[c]); op.append(pi[p]); om.append(m); info_obs = s if info_obs is None else info_obs
            if not np.isclose(s, info_obs, rtol=1e-6): raise ValueError("g2o: static observations with different information")
    pr.update(obs_cam=np.array(oc, np.int32), obs_pt=np.array(op, np.int32), obs_meas=np.array(om, np.float64).reshape(-1, 3), info_obs=info_obs if info_obs is not None else 1.0)
    oi, oj, oT, si, sj = [], [], [], [], []
    info_odo = info_smooth = None
    for i, j, m, inf in e_se3:
        s = _uniform_info(inf, 6, "pose-pose")
        if role[i] != role[j]: raise ValueError("g2o: EDGE_SE3:QUAT between a camera and an object motion")
        if role[i] == "cam":
            oi.append(ci[i]); oj.append(ci[j]); oT.append(_se3(m)); info_odo = s if info_odo is None else info_odo
            if not np.isclose(s, info_odo, rtol=1e-6): raise ValueError("g2o: odometry edges with different information")
        else:
            if not np.allclose(_se3(m), np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0]), atol=1e-9): raise ValueError("g2o: motion smoothness edge with a non-identity measurement")
            si.append(hi[i]); sj.append(hi[j]); info_smooth = s if info_smooth is None else info_smooth
    pr.update(odo_i=np.array(oi, np.int32), odo_j=np.array(oj, np.int32), odo_T=np.array(oT, np.float64).reshape(-1, 12), info_odo=info_odo if info_odo is not None else 1.0)
    ids = dict(cam=cams, pt=pts, H=Hs, dyn=dyns)
    if not Hs and not dyns: return pr, None, ids
    for k in dyns:
        if k not in dyn_of: raise ValueError("g2o: dynamic point %d has no observation" % k)
    info_tern = None; tp, tc, th = [], [], []
    for a, b, h, inf in e_mot:
        s = _uniform_info(inf, 3, "motion"); info_tern = s if info_tern is None else info_tern
        tp.append(di[a]); tc.append(di[b]); th.append(hi[h])
    dy = dict(n_H=len(Hs), H_T=np.array([_se3(se3[k]) for k in Hs]).reshape(-1, 12), n_dyn=len(dyns), dyn_xyz=np.array([xyz[k] for k in dyns], np.float64).reshape(-1, 3),
              dyn_cam=np.array([dyn_of[k][0] for k in dyns], np.int32), dyn_meas=np.array([dyn_of[k][1] for k in dyns], np.float64).reshape(-1, 3),
              n_tern=len(tp), tern_prev=np.array(tp, np.int32), tern_cur=np.array(tc, np.int32), tern_H=np.array(th, np.int32),
              n_smooth=len(si), sm_i=np.array(si, np.int32), sm_j=np.array(sj, np.int32),
              info_dyn=info_dyn if info_dyn is not None else 1.0, info_tern=info_tern if info_tern is not None else 1.0, info_smooth=info_smooth if info_smooth is not None else 1.0,
              huber_dyn=huber, huber_tern=huber, huber_smooth=huber)
    return pr, dy, ids


def write_g2o(path, pr, dy=None, ids=None):
    """Inverse of read_g2o (same tags, 9 significant digits like the facade's dump).  ids: file ids per vertex (default: cameras, landmarks,
    motions, dynamic points numbered 1.. in that order)."""
    n_cam, n_pt = int(pr["n_cam"]), int(pr["n_pt"]); n_H = int(dy["n_H"]) if dy else 0; n_dyn = int(dy["n_dyn"]) if dy else 0
    if ids is None:
        base = 1; ids = {}
        for name, n in (("cam", n_cam), ("pt", n_pt), ("H", n_H), ("dyn", n_dyn)): ids[name] = list(range(base, base + n)); base += n
    cam_T = np.asarray(pr["cam_T"], np.float64).reshape(-1, 3, 4); pt = np.asarray(pr["pt_xyz"], np.float64).reshape(-1, 3)
    g = lambda x: "%.9g" % x
    def se3(T): T = np.asarray(T, np.float64).reshape(3, 4); return " ".join(g(x) for x in list(T[:, 3]) + list(rot_to_quat(T[:, :3])))
    def info(n, s): return " ".join(g(s if i == j else 0.0) for i in range(n) for j in range(i, n))
    verts = [(ids["cam"][i], "VERTEX_SE3:QUAT %d %s" % (ids["cam"][i], se3(cam_T[i]))) for i in range(n_cam)]
    verts += [(ids["pt"][i], "VERTEX_TRACKXYZ %d %s" % (ids["pt"][i], " ".join(g(x) for x in pt[i]))) for i in range(n_pt)]
    if dy:
        H_T = np.asarray(dy["H_T"], np.float64).reshape(-1, 3, 4); dx = np.asarray(dy["dyn_xyz"], np.float64).reshape(-1, 3)
        verts += [(ids["H"][i], "VERTEX_SE3:QUAT %d %s" % (ids["H"][i], se3(H_T[i]))) for i in range(n_H)]
        verts += [(ids["dyn"][i], "VERTEX_TRACKXYZ %d %s" % (ids["dyn"][i], " ".join(g(x) for x in dx[i]))) for i in range(n_dyn)]
    with open(path, "w") as f:
        f.write("PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1 \n")
        for _, line in sorted(verts): f.write(line + " \n")
        if pr["prior_cam"] >= 0: f.write("EDGE_SE3_PRIOR %d 0 %s %s \n" % (ids["cam"][pr["prior_cam"]], se3(pr["prior_T"]), info(6, pr["info_prior"])))
        oT = np.asarray(pr["odo_T"], np.float64).reshape(-1, 12)
        for k in range(len(pr["odo_i"])): f.write("EDGE_SE3:QUAT %d %d %s %s \n" % (ids["cam"][pr["odo_i"][k]], ids["cam"][pr["odo_j"][k]], se3(oT[k]), info(6, pr["info_odo"])))
        om = np.asarray(pr["obs_meas"], np.float64).reshape(-1, 3)
        for k in range(len(pr["obs_cam"])): f.write("EDGE_SE3_TRACKXYZ %d %d 0 %s %s \n" % (ids["cam"][pr["obs_cam"][k]], ids["pt"][pr["obs_pt"][k]], " ".join(g(x) for x in om[k]), info(3, pr["info_obs"])))
        if dy:
            I = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0])
            for k in range(int(dy["n_smooth"])): f.write("EDGE_SE3:QUAT %d %d %s %s \n" % (ids["H"][dy["sm_i"][k]], ids["H"][dy["sm_j"][k]], se3(I), info(6, dy["info_smooth"])))
            dm = np.asarray(dy["dyn_meas"], np.float64).reshape(-1, 3)
            for k in range(n_dyn): f.write("EDGE_SE3_TRACKXYZ %d %d 0 %s %s \n" % (ids["cam"][dy["dyn_cam"][k]], ids["dyn"][k], " ".join(g(x) for x in dm[k]), info(3, dy["info_dyn"])))
            for k in range(int(dy["n_tern"])): f.write("EDGE_SE3_MOTION %d %d %d 0 0 0 %s \n" % (ids["dyn"][dy["tern_prev"][k]], ids["dyn"][dy["tern_cur"][k]], ids["H"][dy["tern_H"][k]], info(3, dy["info_tern"])))
