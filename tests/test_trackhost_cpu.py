"""CPU (no GPU needed): the tracker's host-side bookkeeping stages of the product (vido-slam_amd/csrc/trackhost.cpp through the C-ABI, what the facade's
Tracking::RenewFrameInfo / DynObjTracking / Frame::UndistortKeyPoints / Map::UpdateTracklets run) against the oracle's literal restatements of the reference loops
(oracle/trackhost_oracle.c: Tracking.cc:1670-1912, 2514-2720, 2959-3289, Frame.cc:603-633).  SURVEY.md rows A11, A21, A22, A23.

What differs between the two sides — and is therefore what these tests pin — : the "is the sample within 1 px of a kept inlier" test (grid hash vs the reference's scan
over the whole kept set, O(N*M)), label look-ups (binary search vs linear search), the tracklet store (incremental vs full rebuild every frame)."""
import ctypes as C
import numpy as np
import pytest

import vido_slam_amd as V
from vido_slam_amd import host

pytestmark = pytest.mark.usefixtures("box")          # host + GPU box (tests/conftest.py): the driver's -m gpu run executes these rows too

KAIST_K = (816.402, 817.38, 608.2658, 266.688)                      # src/config/kaist_config.yaml:24-33
KAIST_DIST = (-0.05004, 0.120012, -0.0006259, -0.00118, -0.063505)


def make_maps(rng, h, w, n_blobs=6, zero_flow_frac=0.03):
    mask = np.zeros((h, w), np.int32)
    for k in range(n_blobs):
        y0, x0 = rng.randint(0, h - 60), rng.randint(0, w - 90)
        mask[y0:y0 + rng.randint(30, 60), x0:x0 + rng.randint(40, 90)] = k % 4 + 1            # labels repeat: two blobs may carry one label
    depth = rng.uniform(-2, 60, (h, w)).astype(np.float32)                                     # some <= 0, some > 40, some > 25
    depth[mask > 0] = rng.uniform(3, 30, int((mask > 0).sum())).astype(np.float32)
    flow = rng.uniform(-6, 6, (h, w, 2)).astype(np.float32)
    z = rng.rand(h, w) < zero_flow_frac
    flow[z, 0] = 0
    flow[rng.rand(h, w) < zero_flow_frac, 1] = 0
    return mask, depth, flow


def test_undistort_points_kaist(oracle):
    """A11 with the KAIST intrinsics (k1 = -0.05004, 1280 x 560): product == oracle bit for bit, and the result really is the inverse of the Brown model."""
    rng = np.random.RandomState(1)
    xy = np.stack([rng.uniform(0, 1280, 4000), rng.uniform(0, 560, 4000)], 1).astype(np.float32)
    got = V.undistort_points(xy, KAIST_K, KAIST_DIST)
    ref = oracle.undistort_points(xy, KAIST_K, KAIST_DIST)
    assert np.array_equal(got, ref)
    # independent check in numpy float64: distorting the undistorted points gives the input back (the 5-step fixed point converges to ~1e-4 px at these coefficients)
    fx, fy, cx, cy = KAIST_K; k1, k2, p1, p2, k3 = KAIST_DIST
    x = (got[:, 0].astype(np.float64) - cx) / fx; y = (got[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y; rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([xd * fx + cx, yd * fy + cy], 1)
    assert np.abs(back - xy).max() < 2e-3
    assert np.abs(got - xy).max() > 1.0                                  # and it is not a no-op: corners move by more than a pixel
    # k1 == 0: keys are copied (Frame.cc:605-609)
    assert np.array_equal(V.undistort_points(xy, KAIST_K, (0.0, 0.1, 0.0, 0.0, 0.0)), xy)


@pytest.mark.parametrize("seed,max_num,size", [(3, 3000, (480, 640)), (4, 600, (480, 640)), (5, 40, (480, 640)), (6, 3000, (560, 1280)), (7, 1000, (192, 640))])
def test_renew_static_matches_reference_loops(oracle, seed, max_num, size):
    """A22, static part.  Samples sit on top of kept inliers, within a pixel of them (all eight directions, distances straddling 1.0), on image borders and on cell
    borders of the product's hash grid: the accept / reject decision and therefore the ORDER of the kept list must equal the reference's scan."""
    rng = np.random.RandomState(seed); h, w = size
    mask, depth, flow = make_maps(rng, h, w)
    n_stat = 1800
    stat = np.stack([rng.uniform(-3, w + 3, n_stat), rng.uniform(-3, h + 3, n_stat)], 1).astype(np.float32)
    TM = rng.permutation(n_stat)[:1500].astype(np.int32); TM[rng.rand(len(TM)) < 0.2] = -1
    near = stat[rng.randint(0, n_stat, 700)] + rng.choice([-1.0, -0.75, -0.5, 0.0, 0.5, 0.70710677, 0.75, 1.0, 1.001], (700, 2)).astype(np.float32)
    far = np.stack([rng.uniform(0, w, 1500), rng.uniform(0, h, 1500)], 1).astype(np.float32)
    edge = np.stack([rng.choice([0.0, 0.4, 1.0, w - 1.0, w - 0.5, float(w)], 60), rng.uniform(0, h, 60)], 1).astype(np.float32)
    samples = np.concatenate([near, far, edge, np.floor(far[:200]) + np.float32(0.999)]).astype(np.float32); samples = samples[rng.permutation(len(samples))]
    got = V.renew_static(mask, depth, flow, stat, TM, samples, max_num)
    ref = oracle.renew_static(mask, depth, flow, stat, TM, samples, max_num)
    assert len(got[0]) == len(ref[0]) > 0
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    if max_num >= 3000:
        assert (ref[1] == -1).sum() > 100 and (ref[1] >= 0).sum() > 100       # both stages contributed


@pytest.mark.parametrize("seed,max_obj", [(11, 800), (12, 120), (13, 30)])
def test_renew_objects_matches_reference_loops(oracle, seed, max_obj):
    """A22, object part: tracked objects' inliers snapped to integer pixels, per-object top-up by semantic label in stride-15 passes with the 1-px test, failed
    objects skipped, samples of labels no live object owns appended with label -2."""
    rng = np.random.RandomState(seed); h, w = 480, 640
    mask, depth, flow = make_maps(rng, h, w, n_blobs=7)
    ys, xs = np.nonzero(mask[::4, ::4])
    tmp_xy = np.stack([xs * 4.0, ys * 4.0], 1).astype(np.float32)       # this frame's dense samples (Frame.cc:184-211)
    tmp_sem = mask[::4, ::4][ys, xs].astype(np.int32)
    nt = len(tmp_sem)
    tmp_depth = depth[::4, ::4][ys, xs]; tmp_flow = flow[::4, ::4][ys, xs]; tmp_corr = tmp_xy + tmp_flow
    n_pts = 2500
    obj_xy = (tmp_xy[rng.randint(0, nt, n_pts)] + rng.uniform(-0.9, 0.9, (n_pts, 2))).astype(np.float32)
    obj_xy[:40] = np.stack([rng.uniform(-2, w + 2, 40), rng.uniform(-2, h + 2, 40)], 1)
    obj_label = rng.randint(-1, 6, n_pts).astype(np.int32)
    sets = [rng.permutation(n_pts)[:rng.randint(20, 400)].astype(np.int32) for _ in range(4)]
    obj_stat = np.array([1, 0, 1, 1], np.uint8); sem_position = np.array([1, 2, 3, 3], np.int32); mod_label = np.array([4, 5, 6, 9], np.int32)
    args = (mask, depth, flow, obj_xy, obj_label, sets, obj_stat, sem_position, mod_label, tmp_xy, tmp_depth, tmp_sem, tmp_flow, tmp_corr, max_obj)
    got = V.renew_objects(*args); ref = oracle.renew_objects(*args)
    assert len(ref["keys"]) > 200
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert (ref["label"] == -2).any() and (ref["inlier"] >= 0).any()
    if max_obj >= 800:
        assert ((ref["inlier"] == -1) & (ref["label"] >= 0)).any()         # the per-object top-up contributed


@pytest.mark.parametrize("seed,f_id,max_id", [(21, 1, 7), (22, 5, 4), (23, 5, 1), (24, 9, 12)])
def test_dyn_obj_tracking_matches_reference(oracle, seed, f_id, max_id):
    """A21: border rejection, static / far / small decisions, identity from the previous frame's objects (dominant last label, ties to the lower label), new ids."""
    rng = np.random.RandomState(seed); rows, cols = 480, 640
    n = 4200
    sem = rng.choice([1, 2, 3, 5, 8, 9], n, p=[0.3, 0.25, 0.2, 0.15, 0.06, 0.04]).astype(np.int32)
    xy = np.stack([rng.uniform(0, cols, n), rng.uniform(0, rows, n)], 1).astype(np.float32)
    xy[sem == 5, 0] = rng.uniform(0, 25, int((sem == 5).sum()))                          # label 5 hugs the left border: dropped
    depth = rng.uniform(4, 20, n).astype(np.float32); depth[sem == 3] += 30               # label 3 is far
    f3 = rng.normal(0, 0.5, (n, 3)).astype(np.float32); f3[sem == 2] *= 0.05               # label 2 barely moves: static
    obj_label = np.full(n, -2, np.int32); obj_label[rng.rand(n) < 0.1] = -1
    last_sem = sem.copy(); flip = rng.rand(n) < 0.3; last_sem[flip] = rng.choice([1, 2, 9], int(flip.sum()))
    eq = np.nonzero(sem == 9)[0]; last_sem[eq[: len(eq) // 2]] = 1; last_sem[eq[len(eq) // 2: 2 * (len(eq) // 2)]] = 9        # exact tie 1 vs 9 -> 1
    last_pos = np.array([1, 9, 2, 1], np.int32); last_stat = np.array([0, 1, 1, 1], np.uint8); last_mod = np.array([3, 6, 2, 5], np.int32)
    args = (sem, obj_label, xy, depth, f3, last_sem, last_pos, last_stat, last_mod, rows, cols, 0.12, 0.3, 25.0, f_id, max_id)
    got = V.dyn_obj_tracking(*args); ref = oracle.dyn_obj_tracking(*args)
    assert len(ref["objects"]) >= 1 and got["max_id"] == ref["max_id"]
    assert np.array_equal(got["obj_label"], ref["obj_label"]) and np.array_equal(got["mod_label"], ref["mod_label"]) and np.array_equal(got["sem_position"], ref["sem_position"])
    assert len(got["objects"]) == len(ref["objects"]) and all(np.array_equal(a, b) for a, b in zip(got["objects"], ref["objects"]))
    assert set(np.unique(ref["obj_label"])) >= {-1, 0}


def _incremental(rows, labels, n_feat0):
    lib = host.load_library()
    n_rows = len(rows); row_n = np.array([len(r) for r in rows], np.int32); row_off = np.zeros(n_rows + 1, np.int32); row_off[1:] = np.cumsum(row_n)
    TM = np.ascontiguousarray(np.concatenate(rows), np.int32)
    lab = np.ascontiguousarray(np.concatenate(labels), np.int32) if labels is not None else None
    cap_t = int((TM >= 0).sum()) + 1; cap_p = 2 * cap_t + 2; nfeat = n_feat0 + int(row_n.sum())
    off = np.zeros(cap_t + 1, np.int32); pairs = np.zeros((cap_p, 2), np.int32); oid = np.zeros(cap_t, np.int32); ot = np.zeros(nfeat, np.int32); op = np.zeros(nfeat, np.int32); nt = C.c_int32()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.vido_tracklets_incremental(n_rows, p(row_off), p(row_n), p(TM), p(lab) if lab is not None else None, n_feat0, p(off), p(pairs), p(oid), p(ot), p(op), cap_t, cap_p, C.byref(nt))
    assert rc == 0
    nt = nt.value
    return [[tuple(int(v) for v in pr) for pr in pairs[off[t]:off[t + 1]]] for t in range(nt)], oid[:nt].copy(), ot, op


@pytest.mark.parametrize("seed,dyn", [(31, False), (32, True), (33, False)])
def test_incremental_tracklets_equal_the_full_rebuild(oracle, seed, dyn):
    """A23: Map::UpdateTracklets fed one frame at a time == GetStaticTrack / GetDynamicTrackNew rebuilt from frame 0 (Tracking.cc:2514-2720), incl. the object id of
    dynamic tracklets; and the per-feature owner tables say what the optimisers derive from the full list (a feature belongs to the highest-numbered tracklet of
    length >= 3 that contains it)."""
    rng = np.random.RandomState(seed)
    n_frames = 14; n0 = 60
    rows, labels, prev_n = [], [], n0
    for f in range(1, n_frames):
        n = rng.randint(30, 80)
        r = rng.randint(-1, prev_n, n).astype(np.int32)             # several features may claim one predecessor (the reference allows it: the tracklet forks)
        r[rng.rand(n) < 0.35] = -1
        rows.append(r); labels.append(rng.randint(1, 5, n).astype(np.int32)); prev_n = n
    got, oid, ot, op = _incremental(rows, labels if dyn else None, n0)
    ref, roid = oracle.tracklets(rows, labels if dyn else None)
    assert len(got) == len(ref) > 50 and got == ref
    if dyn:
        assert np.array_equal(oid, roid)
    # owner tables from the definition
    sizes = [n0] + [len(r) for r in rows]; base = np.concatenate([[0], np.cumsum(sizes)])
    want_t = np.full(base[-1], -1, np.int32); want_p = np.full(base[-1], -1, np.int32)
    for t, tr in enumerate(ref):
        if len(tr) >= 3:
            for k, (f, j) in enumerate(tr):
                if want_t[base[f] + j] <= t:
                    want_t[base[f] + j] = t; want_p[base[f] + j] = k
    assert np.array_equal(ot, want_t) and np.array_equal(op, want_p)


def _cv_rng_gaussian_py(seed):
    """A fresh cv::RNG(seed).gaussian(1.0), written from the published algorithm a second time (numpy float32 where OpenCV computes in float): multiply-with-carry state,
    128-strip ziggurat (Marsaglia & Tsang) on the state's low word BEFORE the step (modules/core/src/rand.cpp::randn_0_1_32f)."""
    f32 = np.float32
    m1 = 2147483648.0; dn = 3.442619855899; tn = dn; vn = 9.91256303526217e-3
    q = vn / np.exp(-0.5 * dn * dn)
    kn = [0] * 128; wn = [f32(0)] * 128; fn = [f32(0)] * 128
    kn[0] = int((dn / q) * m1) & 0xffffffff; kn[1] = 0
    wn[0] = f32(q / m1); wn[127] = f32(dn / m1); fn[0] = f32(1.0); fn[127] = f32(np.exp(-0.5 * dn * dn))
    for i in range(126, 0, -1):
        dn = np.sqrt(-2.0 * np.log(vn / dn + np.exp(-0.5 * dn * dn)))
        kn[i + 1] = int((dn / tn) * m1) & 0xffffffff; tn = dn
        fn[i] = f32(np.exp(-0.5 * dn * dn)); wn[i] = f32(dn / m1)
    st = seed if seed else 0xffffffff
    def step(s): return ((s & 0xffffffff) * 4164903690 + (s >> 32)) & 0xffffffffffffffff
    rng_flt = f32(2.3283064365386962890625e-10)
    while True:
        lo = st & 0xffffffff; hz = lo - (1 << 32) if lo >= (1 << 31) else lo
        st = step(st); iz = hz & 127
        x = f32(f32(hz) * wn[iz])
        if abs(hz) < kn[iz]:
            return float(x)
        if iz == 0:
            while True:
                x = f32(f32(st & 0xffffffff) * rng_flt); st = step(st)
                y = f32(f32(st & 0xffffffff) * rng_flt); st = step(st)
                x = f32(-np.log(np.float64(x) + np.finfo(np.float32).tiny) * 0.2904764); y = f32(-np.log(np.float64(y) + np.finfo(np.float32).tiny))
                if not (f32(y + y) < f32(x * x)):
                    break
            return float(f32(3.442620) + x) if hz > 0 else float(-f32(3.442620) - x)
        y = f32(f32(st & 0xffffffff) * rng_flt); st = step(st)
        if f32(fn[iz] + y * f32(fn[iz - 1] - fn[iz])) < f32(np.exp(-0.5 * np.float64(x) * np.float64(x))):
            return float(x)


def test_depth_noise_is_one_draw_of_a_fresh_rng_per_call():
    """vido_depth_noise (Frame.cc:711-716: `cv::RNG rng((unsigned)time(NULL)); z = z + rng.gaussian(z*z/(725*0.5)*0.15)`): a deterministic function of (z, seed) — the same
    standard-normal draw for every point of one second — against a second restatement of cv::RNG + the ziggurat; accepted strips, wedges and the two seeds that land in the
    base strip (seed & 127 == 0)."""
    lib = V.load_library()
    lib.vido_depth_noise.restype = C.c_float; lib.vido_depth_noise.argtypes = [C.c_float, C.c_uint]
    seeds = [1, 2, 127, 128, 256, 1758900000, 1758900001, 1758900127, 1758900096, 0x7fffffff, 0x80000000, 0xfffffffe, 0x9e3779b9, 4096, 1 << 20]
    for seed in seeds:
        g = _cv_rng_gaussian_py(seed)
        for z in (0.5, 7.25, 31.0, 80.0):
            zf = np.float32(z)
            want = np.float32(np.float64(zf) + np.float64(np.float32(g)) * (np.float64(np.float32(zf * zf)) / (725 * 0.5) * 0.15))
            got = lib.vido_depth_noise(float(zf), seed)
            assert got == want, (seed, z, got, want, g)
    # one seed = one offset: the relative perturbation grows with z (sigma = 0.15 z^2 / 362.5), and a second's worth of calls shares it
    a = [lib.vido_depth_noise(z, 1758900000) - z for z in (5.0, 10.0, 20.0)]
    assert abs(a[1] / a[0] - 4.0) < 1e-3 and abs(a[2] / a[0] - 16.0) < 1e-3


@pytest.mark.parametrize("n,nbins,seed", [(0, 5, 0), (1, 1, 1), (1000, 7, 2), (50000, 1024, 3), (200000, 1025, 4), (300000, 100000, 5), (4096, 70000, 6)])
def test_stable_rank_of_the_ba_setup_equals_a_stable_argsort(n, nbins, seed):
    """vido_debug_stable_rank (csrc/ba.hip: par_counting_rank / par_counting_rank_large on the set-up's host pool): what sorts a global bundle adjustment's observation
    list by camera (500 bins, one level) and by landmark (100 k bins, two levels).  pos[i] must be the rank of i in a STABLE sort by key, bin_start the first rank of each
    key — the order of a landmark's slots (ascending camera) depends on the stability."""
    import ctypes as C
    from vido_slam_amd.host import load_library
    lib = load_library()
    rng = np.random.default_rng(seed)
    keys = (rng.integers(0, nbins, n) if n else np.zeros(0)).astype(np.int32)
    if n > 10:
        keys[: n // 3] = keys[0]                                   # a heavy bin and many empty ones
    pos = np.full(n, -1, np.int32); bs = np.full(nbins + 1, -1, np.int32)
    assert lib.vido_debug_stable_rank(C.c_void_p(keys.ctypes.data), n, nbins, C.c_void_p(pos.ctypes.data), C.c_void_p(bs.ctypes.data)) == 0
    order = np.argsort(keys, kind="stable")
    ref = np.empty(n, np.int32); ref[order] = np.arange(n, dtype=np.int32)
    assert np.array_equal(pos, ref)
    assert np.array_equal(bs, np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=nbins))]).astype(np.int32))
    bad = np.array([0, nbins], np.int32)
    assert lib.vido_debug_stable_rank(C.c_void_p(bad.ctypes.data), 2, nbins, C.c_void_p(pos.ctypes.data), C.c_void_p(bs.ctypes.data)) != 0      # a key outside [0, nbins)
