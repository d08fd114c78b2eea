"""GPU parity: vido_ba_optimize (LM + point-Schur on the device) vs the CPU oracle (oracle/ba_oracle.c, restating
PartialBatchOptimization / the static part of FullBatchOptimization on g2o's LM).  Tolerance: camera SE(3) and
landmarks within 1e-4 relative (BASELINE.json north_star); FP64 atomics make the summation order free, so the
comparison is not bitwise."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def ctx(vido):
    return vido.Context(width=640, height=480, max_batch=1)


@pytest.mark.parametrize("kw", [
    dict(n_cam=20, n_pt=2000, kind="local", seed=7),                       # BASELINE configs[3] static graph (LDS path)
    dict(n_cam=20, n_pt=600, kind="local", seed=8, with_prior=False),      # sliding window: no prior, gauge free
    dict(n_cam=5, n_pt=120, kind="local", seed=9),
    dict(n_cam=60, n_pt=3000, kind="global", track_len=10, seed=11),       # HBM-atomics path + blocked Cholesky
    dict(n_cam=23, n_pt=900, kind="global", track_len=7, seed=12),         # n6 = 138: just above the LDS limit
    dict(n_cam=90, n_pt=1500, kind="global", track_len=30, seed=13),       # 30-frame tracks: band too wide for the LDS window -> supernodal k_chol_band6s
    dict(n_cam=80, n_pt=2500, kind="global", track_len=14, seed=14),       # half-bandwidth 89: block cyclic reduction with 96-unknown superblocks (k_bcr_elim<96>); the 60-camera case runs k_bcr_elim<66>
])
def test_ba_matches_oracle(vido, oracle, ctx, kw):
    pr = vido.problems.synth_ba_problem(**kw)
    ref = oracle.ba_optimize(pr)
    got = vido.ba_optimize(ctx, pr)
    assert got["iterations"] == ref["iterations"] and got["lm_trials"] == ref["lm_trials"], (got["iterations"], ref["iterations"])
    assert abs(got["chi2_initial"] - ref["chi2_initial"]) <= 1e-9 * ref["chi2_initial"]
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    assert rel(got["cam_T"], ref["cam_T"]) < RTOL
    assert rel(got["pt_xyz"], ref["pt_xyz"]) < RTOL
    # and the optimisation did its job
    assert got["chi2_final"] < 0.7 * got["chi2_initial"]
    assert np.abs(got["cam_T"] - pr["cam_true"]).max() < 0.5 * np.abs(pr["cam_T"] - pr["cam_true"]).max()


def test_zero_noise_problem_stays_at_truth(vido, ctx):
    pr = vido.problems.synth_ba_problem(n_cam=8, n_pt=300, seed=3, obs_noise=0.0, pose_noise=0.0)
    pr["pt_xyz"] = pr["pt_true"].copy(); pr["cam_T"] = pr["cam_true"].copy()
    pr["odo_T"] = np.stack([vido.problems._iso(np.linalg.inv(np.vstack([pr["cam_true"][i], [0, 0, 0, 1]])) @ np.vstack([pr["cam_true"][i + 1], [0, 0, 0, 1]])) for i in range(7)])
    got = vido.ba_optimize(ctx, pr)
    assert got["chi2_final"] < 1e-12 and np.abs(got["cam_T"] - pr["cam_true"]).max() < 1e-9


def test_sharded_single_process_equals_unsharded(vido, ctx, oracle):
    """The landmark-shard code path with a local 'all-reduce' over two sequentially evaluated shards is exercised
    on the CPU by tests/test_ba_shard_cpu.py (gloo); here: shard covering everything == default."""
    pr = vido.problems.synth_ba_problem(n_cam=12, n_pt=500, seed=5)
    a = vido.ba_optimize(ctx, pr)
    b = vido.ba_optimize(ctx, pr, shard=(0, pr["n_pt"]))
    assert np.array_equal(a["cam_T"], b["cam_T"]) or rel(a["cam_T"], b["cam_T"]) < 1e-9


def _rank_hooks(vido, world=2):
    """An in-place 'all-reduce' between `world` vido_ba_optimize calls running in as many threads on ONE GPU (one context = one stream each): every rank brings its
    device buffer to the host, the ranks meet at a barrier, each writes the sum / max back.  What the RCCL hook does over xGMI, without a second GPU."""
    import threading
    import torch
    from vido_slam_amd.host import ALLREDUCE_FN, _DevBuf
    barrier = threading.Barrier(world, timeout=180); bufs = [None] * world; calls = [[] for _ in range(world)]

    def make(rank):
        def hook(user, ptr, count, op):
            try:
                t = torch.as_tensor(_DevBuf(ptr, count), device="cuda")
                bufs[rank] = t.cpu().numpy().copy()
                barrier.wait()
                s = bufs[0].copy()
                for o in bufs[1:]:                      # fixed rank order: every rank computes the identical sum
                    s = s + o if op == 0 else np.maximum(s, o)
                barrier.wait()
                t.copy_(torch.from_numpy(s)); torch.cuda.synchronize()
                calls[rank].append((int(count), int(op)))
                return 0
            except Exception as e:           # a broken barrier etc.: fail the call instead of hanging the other rank
                print("hook rank %d: %r" % (rank, e)); barrier.abort(); return 1
        return ALLREDUCE_FN(hook)
    return make, calls


@pytest.mark.parametrize("kw,dyn", [
    (dict(n_cam=60, n_pt=3000, kind="global", track_len=10, seed=11), False),     # band layout, pose-block Cholesky
    (dict(n_cam=23, n_pt=900, kind="global", track_len=7, seed=12), False),
    (dict(n_cam=20, n_pt=2000, kind="local", seed=7), False),                     # LDS-resident reduced system, sharded all the same
    (dict(n_cam=48, n_pt=1500, kind="global", track_len=8, seed=15), True),       # + object factors (rank 0 owns them)
])
def test_two_shards_on_one_gpu_equal_unsharded_and_oracle(vido, oracle, kw, dyn):
    """The product's sharded path — rank 1 (no camera-camera factors, add_cam = 0), the three all-reduces per LM trial, landmark ranges — executed for real:
    vido_ba_optimize as rank 0 and rank 1 of world 2, concurrently, each on its own context, exchanging through the hook above."""
    import threading
    pr = vido.problems.synth_ba_problem(**kw); pr["max_iters"] = 8
    dy = vido.problems.synth_ba_dynamic(pr, n_obj=2, pts_per_obj=40, seed=16, max_len=6) if dyn else None
    ctxs = [vido.Context(width=640, height=480, max_batch=1) for _ in range(3)]
    ref = vido.ba_optimize(ctxs[2], pr, dynamic=dy)
    shards = vido.landmark_shards(pr["obs_pt"], pr["n_pt"], 2)
    assert shards[0][1] == shards[1][0] and 0 < shards[0][1] < pr["n_pt"]
    make, calls = _rank_hooks(vido, 2)
    out = [None, None]; err = [None, None]

    def run(rank):
        try:
            out[rank] = vido.ba_optimize(ctxs[rank], pr, rank=rank, world=2, shard=shards[rank], allreduce=make(rank), dynamic=dy)
        except Exception as e:
            err[rank] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert err == [None, None], err
    a, b = out
    assert len(calls[0]) == len(calls[1]) and len(calls[0]) >= 3 * a["lm_trials"]       # the collectives really ran, the same sequence on both ranks
    assert np.array_equal(a["cam_T"], b["cam_T"])                                          # replicated reduced solve: bit-identical poses on both ranks
    assert a["iterations"] == b["iterations"] == ref["iterations"] and a["lm_trials"] == b["lm_trials"] == ref["lm_trials"]
    assert rel(a["cam_T"], ref["cam_T"]) < 1e-8 and abs(a["chi2_final"] - ref["chi2_final"]) <= 1e-8 * ref["chi2_final"]
    lo, hi = shards[0]
    assert rel(a["pt_xyz"][lo:hi], ref["pt_xyz"][lo:hi]) < 1e-8 and rel(b["pt_xyz"][hi:], ref["pt_xyz"][hi:]) < 1e-8    # every rank refines (and returns) its own landmarks
    if dyn:
        assert rel(a["H_T"], ref["H_T"]) < 1e-8 and rel(a["dyn_xyz"], ref["dyn_xyz"]) < 1e-8
    else:
        o = oracle.ba_optimize(pr)
        assert a["iterations"] == o["iterations"] and rel(a["cam_T"], o["cam_T"]) < RTOL
    for c in ctxs:
        c.close()


def _run_sharded(vido, pr, world, dy=None):
    """vido_ba_optimize as ranks 0..world-1 concurrently on ONE GPU (one context and thread per rank), exchanging through _rank_hooks."""
    import threading
    ctxs = [vido.Context(width=640, height=480, max_batch=1) for _ in range(world)]
    shards = vido.landmark_shards(pr["obs_pt"], pr["n_pt"], world)
    make, calls = _rank_hooks(vido, world)
    out = [None] * world; err = [None] * world

    def run(rank):
        try:
            out[rank] = vido.ba_optimize(ctxs[rank], pr, rank=rank, world=world, shard=shards[rank], allreduce=make(rank), dynamic=dy)
        except Exception as e:
            err[rank] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    for c in ctxs:
        c.close()
    assert err == [None] * world, err
    return out, shards, calls


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_shards_on_one_gpu_equal_unsharded(vido, ctx, world):
    """The 8-GPU partitioning of BASELINE configs[4] (landmark ranges balanced by observation count, rank 0 owns the camera-camera factors) executed for real with 4 and 8
    ranks on one MI355X, 120 k observations: every rank ends on the same poses as the unsharded solve, its own landmark range equals the unsharded one."""
    pr = vido.problems.synth_ba_problem(n_cam=120, n_pt=12000, kind="global", track_len=10, seed=17); pr["max_iters"] = 6
    assert len(pr["obs_cam"]) >= 100000
    ref = vido.ba_optimize(ctx, pr)
    out, shards, calls = _run_sharded(vido, pr, world)
    assert all(len(c) == len(calls[0]) for c in calls) and len(calls[0]) >= 3 * ref["lm_trials"]
    for r, o in enumerate(out):
        assert (o["iterations"], o["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
        assert np.array_equal(o["cam_T"], out[0]["cam_T"])
        assert rel(o["cam_T"], ref["cam_T"]) < 1e-8 and abs(o["chi2_final"] - ref["chi2_final"]) <= 1e-8 * ref["chi2_final"]
        lo, hi = shards[r]
        assert rel(o["pt_xyz"][lo:hi], ref["pt_xyz"][lo:hi]) < 1e-8


def test_configs4_full_size_500_keyframes_100k_landmarks(vido, oracle, ctx):
    """BASELINE configs[4] at its defining size on one GPU: 500 keyframes x 100 000 landmarks, ~1 M observations, against the oracle
    (oracle/ba_oracle.c: point-Schur + dense LDL^T of the 3 000-unknown reduced system, ~5 s per LM iteration on one host core), plus the size-independent properties:
    chi2 never increases over the accepted iterations, the estimate moves towards the ground truth, two shards on one GPU reproduce the unsharded solve."""
    pr = vido.problems.synth_ba_problem(n_cam=500, n_pt=100000, kind="global", track_len=10, seed=11)
    assert pr["n_cam"] == 500 and pr["n_pt"] == 100000 and len(pr["obs_cam"]) > 900000
    pr["max_iters"] = 4
    ref = oracle.ba_optimize(pr)
    got = vido.ba_optimize(ctx, pr)
    assert (got["iterations"], got["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
    assert abs(got["chi2_initial"] - ref["chi2_initial"]) <= 1e-9 * ref["chi2_initial"]
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    assert rel(got["cam_T"], ref["cam_T"]) < RTOL and rel(got["pt_xyz"], ref["pt_xyz"]) < RTOL
    # monotone chi2: one more iteration never ends higher
    chi = [got["chi2_initial"]]
    for k in (1, 2, 3, 4):
        q = dict(pr); q["max_iters"] = k
        chi.append(vido.ba_optimize(ctx, q)["chi2_final"])
    assert all(b <= a * (1 + 1e-12) for a, b in zip(chi, chi[1:])), chi
    assert got["chi2_final"] < 0.5 * got["chi2_initial"]
    # towards the truth: mean camera position error drops by more than half (the gauge is fixed by the prior on camera 0)
    e0 = np.linalg.norm(pr["cam_T"][:, :, 3] - pr["cam_true"][:, :, 3], axis=1).mean(); e1 = np.linalg.norm(got["cam_T"][:, :, 3] - pr["cam_true"][:, :, 3], axis=1).mean()
    assert e1 < 0.5 * e0, (e0, e1)
    out, shards, calls = _run_sharded(vido, pr, 2)
    for r, o in enumerate(out):
        assert (o["iterations"], o["lm_trials"]) == (got["iterations"], got["lm_trials"]) and rel(o["cam_T"], got["cam_T"]) < 1e-8
        lo, hi = shards[r]
        assert rel(o["pt_xyz"][lo:hi], got["pt_xyz"][lo:hi]) < 1e-8


@pytest.mark.parametrize("n_cam,n_pt", [(60, 3000), (300, 30000)])
def test_observation_order_does_not_matter(vido, ctx, n_cam, n_pt):
    """The set-up sorts the observation list by camera — and skips the sort when the list arrives in camera order, as a SLAM map's does (csrc/ba.hip `in_order`).  The same
    graph with its observations SHUFFLED (the sort runs; above 200 k observations on the host pool) and in landmark-major order must give the same solve; the slot order of a
    landmark (ascending camera) is what the stable sorts guarantee, so the results agree to rounding of the order-independent sums."""
    pr = vido.problems.synth_ba_problem(n_cam=n_cam, n_pt=n_pt, kind="global", track_len=10, seed=23); pr["max_iters"] = 4
    assert np.all(np.diff(pr["obs_cam"]) >= 0)                      # the generator emits camera-major lists: the in-order path
    ref = vido.ba_optimize(ctx, pr)
    rng = np.random.default_rng(5)
    for order in (rng.permutation(len(pr["obs_cam"])), np.lexsort((pr["obs_cam"], pr["obs_pt"]))):
        q = dict(pr); q["obs_cam"] = np.ascontiguousarray(pr["obs_cam"][order]); q["obs_pt"] = np.ascontiguousarray(pr["obs_pt"][order]); q["obs_meas"] = np.ascontiguousarray(pr["obs_meas"][order])
        got = vido.ba_optimize(ctx, q)
        assert (got["iterations"], got["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
        assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-9 * ref["chi2_final"]
        assert rel(got["cam_T"], ref["cam_T"]) < 1e-9 and rel(got["pt_xyz"], ref["pt_xyz"]) < 1e-9


def test_malformed_problem_is_rejected(vido, ctx):
    pr = vido.problems.synth_ba_problem(n_cam=4, n_pt=50, seed=1)
    pr["obs_cam"] = pr["obs_cam"].copy(); pr["obs_cam"][0] = 99
    with pytest.raises(vido.VidoError):
        vido.ba_optimize(ctx, pr)


def test_tracks_longer_than_64_frames_and_duplicate_observations(vido, oracle, ctx):
    """FullBatchOptimization has no track-length limit (Optimizer.cc:1235ff): landmarks seen from 65..120 cameras go through k_ba_schur_long and must give the
    oracle's result; a landmark observed twice from one camera is rejected (the pair kernels assume distinct cameras per landmark)."""
    pr = vido.problems.synth_ba_problem(n_cam=120, n_pt=400, kind="global", track_len=10, seed=31, step=0.05)
    # make 12 landmarks visible from every camera (a long static track) by adding synthetic observations consistent with the true geometry
    rng = np.random.RandomState(3)
    cams = np.stack([np.vstack([c, [0, 0, 0, 1]]) for c in pr["cam_true"]]); inv = np.linalg.inv(cams)
    oc, op, om = [pr["obs_cam"]], [pr["obs_pt"]], [pr["obs_meas"]]
    for l in range(12):
        seen = set(pr["obs_cam"][pr["obs_pt"] == l].tolist())
        for c in range(pr["n_cam"] if l < 6 else 80):
            if c in seen:
                continue
            Xc = inv[c][:3, :3] @ pr["pt_true"][l] + inv[c][:3, 3]
            oc.append(np.array([c], np.int32)); op.append(np.array([l], np.int32)); om.append((Xc + rng.normal(0, 0.01, 3))[None])
    pr["obs_cam"] = np.concatenate(oc); pr["obs_pt"] = np.concatenate(op); pr["obs_meas"] = np.concatenate(om)
    assert np.bincount(pr["obs_pt"]).max() == 120
    pr["max_iters"] = 6
    ref = oracle.ba_optimize(pr)
    got = vido.ba_optimize(ctx, pr)
    assert got["iterations"] == ref["iterations"] and got["lm_trials"] == ref["lm_trials"]
    assert rel(got["cam_T"], ref["cam_T"]) < RTOL and rel(got["pt_xyz"], ref["pt_xyz"]) < RTOL
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    dup = dict(pr); dup["obs_cam"] = np.concatenate([pr["obs_cam"], pr["obs_cam"][:1]]); dup["obs_pt"] = np.concatenate([pr["obs_pt"], pr["obs_pt"][:1]])
    dup["obs_meas"] = np.concatenate([pr["obs_meas"], pr["obs_meas"][:1]])
    with pytest.raises(vido.VidoError):
        vido.ba_optimize(ctx, dup)


def test_block_cyclic_reduction_equals_the_band_cholesky(vido, ctx, monkeypatch):
    """the same problem through k_bcr_* and (VIDO_BA_NO_BCR) through k_chol_band6: same LM path, same result to rounding"""
    pr = vido.problems.synth_ba_problem(n_cam=150, n_pt=9000, kind="global", track_len=10, seed=21); pr["max_iters"] = 6
    a = vido.ba_optimize(ctx, pr)
    monkeypatch.setenv("VIDO_BA_NO_BCR", "1")
    b = vido.ba_optimize(ctx, pr)
    assert a["iterations"] == b["iterations"] and a["lm_trials"] == b["lm_trials"]
    assert abs(a["chi2_final"] - b["chi2_final"]) <= 1e-9 * b["chi2_final"]
    assert rel(a["cam_T"], b["cam_T"]) < 1e-9 and rel(a["pt_xyz"], b["pt_xyz"]) < 1e-9


def test_library_side_rccl_allreduce_world_1(vido):
    """vido_rccl_* (csrc/rccl.cpp): communicator on the context's device, ncclAllReduce enqueued on the context's stream by the sharded code path.  One GPU here,
    so world = 1 (the reduction is the identity): what this pins is the plumbing — librccl resolved at run time, the init sequence, the stream-ordered call without
    host synchronisation, the sharded solve (rank 0 of 1) equal to the plain one.  world > 1 needs one GPU per rank and is the driver's multi-GPU run."""
    c = vido.Context(width=640, height=480, max_batch=1)
    try:
        pr = vido.problems.synth_ba_problem(n_cam=60, n_pt=3000, kind="global", track_len=10, seed=11)
        plain = vido.ba_optimize(c, pr)
        mode = vido.rccl_direct_init(c, 0, 1)
        got = vido.ba_optimize(c, pr, rank=0, world=1, shard=(0, int(pr["n_pt"])), allreduce=mode)
        assert got["iterations"] == plain["iterations"] and got["lm_trials"] == plain["lm_trials"]
        assert rel(got["cam_T"], plain["cam_T"]) < 1e-9 and rel(got["pt_xyz"], plain["pt_xyz"]) < 1e-9
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("share", [0.002, 0.10])
def test_revisited_landmarks_leave_the_camera_window(vido, oracle, ctx, share, capfd, monkeypatch):
    """A map with revisits (loop closures): `share` of the landmarks get two extra observations from cameras 60-70 frames after their track — outside the 16-camera
    window of the matrix-core Schur kernel.  0.2 %: those landmarks go to k_ba_schur_long; 10 %: the wave-per-landmark kernel takes the whole graph (csrc/ba.hip, the 3 % rule).
    Either way the solve equals the oracle's (same LM iteration and trial counts, poses / landmarks to the suite's tolerance)."""
    pr = vido.problems.synth_ba_problem(n_cam=100, n_pt=4000, kind="global", track_len=10, seed=41)
    rng = np.random.RandomState(5); n_pt = pr["n_pt"]
    last = np.zeros(n_pt, np.int64); np.maximum.at(last, pr["obs_pt"], pr["obs_cam"])
    cand = np.nonzero(last + 72 < pr["n_cam"])[0]; pick = rng.choice(cand, int(share * n_pt), replace=False)
    cams = np.stack([np.vstack([c, [0, 0, 0, 1]]) for c in pr["cam_true"]])
    oc, op, om = [pr["obs_cam"]], [pr["obs_pt"]], [pr["obs_meas"]]
    for l in pick:
        for gap in (60, 70):
            c = int(last[l]) + gap; inv = np.linalg.inv(cams[c]); Xc = inv[:3, :3] @ pr["pt_true"][l] + inv[:3, 3]
            oc.append(np.array([c], np.int32)); op.append(np.array([l], np.int32)); om.append((Xc + rng.normal(0, 0.02, 3))[None])
    pr["obs_cam"] = np.concatenate(oc).astype(np.int32); pr["obs_pt"] = np.concatenate(op).astype(np.int32); pr["obs_meas"] = np.concatenate(om)
    pr["max_iters"] = 6
    monkeypatch.setenv("VIDO_BA_VERBOSE", "1")
    ref = oracle.ba_optimize(dict(pr))
    got = vido.ba_optimize(ctx, dict(pr))
    err = capfd.readouterr().err
    import re
    m = re.search(r"outside their chunk's camera window: (\d+) of (\d+) \(([0-9.]+) %\)", err)
    assert m and int(m.group(1)) >= len(pick) and (float(m.group(3)) > 3.0) == (share > 0.05)      # (a few ordinary tracks straddle a window edge as well: ~2 % here)
    assert got["iterations"] == ref["iterations"] and got["lm_trials"] == ref["lm_trials"]
    assert rel(got["cam_T"], ref["cam_T"]) < RTOL and rel(got["pt_xyz"], ref["pt_xyz"]) < RTOL
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"VIDO_BA_PERSIST": "1"}, {"VIDO_BA_NO_FUSED_LOCAL": "1"}, {"VIDO_BA_NO_SPEC": "1"}])
def test_local_window_alternative_drivers_match_the_oracle(env):
    """The local window has four drivers over the same kernels' bodies: the enqueued-ahead solve with the LM state on the device (default), the fused host-driven trial loop
    (VIDO_BA_NO_SPEC=1), the persistent one-launch solver k_ba_local_lm (VIDO_BA_PERSIST=1: opt-in) and round 3's loop (VIDO_BA_NO_FUSED_LOCAL=1).
    The switches are read once per process, so the alternatives run
    the local-window oracle cases (same LM iteration AND trial counts, poses / points to 1e-4) and the facade's resident-window cross-check in a child process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ba_gpu.py"), os.path.join(root, "tests", "test_facade_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "(test_ba_matches_oracle and kw0 or test_ba_matches_oracle and kw1 or test_ba_matches_oracle and kw2 or zero_noise or device_resident_ba_window) and not alternative_drivers"],
                       cwd=root, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "5 passed" in p.stdout, p.stdout[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"VIDO_BCR_SCALAR": "1"}, {"VIDO_BCR_BACK_LEVELS": "1"}, {"VIDO_BA_SCHUR_CHUNK": "64"}, {"VIDO_BA_SCHUR_CHUNK": "512"}, {"VIDO_BA_SCHUR_OLD": "1"}])
def test_global_solver_forms_of_round_6_agree_with_the_older_ones(env, tmp_path):
    """Round 6 changed three things inside a global LM trial: the Schur update of the block cyclic reduction (contraction over four lanes), its back substitution (ONE launch,
    the levels chained by tagged granules: csrc/ba.hip::k_bcr_back_chain) and k_ba_schur_mfma (passes as one prefetching sequence, unit size chosen by the host).  The older
    forms are still selectable by environment (read once per process): a child process solves the same 300-keyframe graph with each, and the results must agree with this
    process' default forms — same LM iteration and trial counts, poses and landmarks to 1e-9 relative (they differ only in the order of a few sums)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import vido_slam_amd as V\n"
            "ctx = V.Context(width=640, height=480, max_batch=1)\n"
            "pr = V.problems.synth_ba_problem(n_cam=300, n_pt=40000, kind='global', track_len=10, seed=5); pr['max_iters'] = 4\n"
            "r = V.ba_optimize(ctx, pr)\n"
            "np.savez(sys.argv[1], cam=r['cam_T'], pt=r['pt_xyz'], it=np.array([r['iterations'], r['lm_trials']]), chi=np.array([r['chi2_initial'], r['chi2_final']]))\n" % root)
    outs = []
    for tag, e in (("default", {}), ("alt", env)):
        f = str(tmp_path / (tag + ".npz"))
        clean = {k: v for k, v in os.environ.items() if not k.startswith("VIDO_BCR_") and k not in ("VIDO_BA_SCHUR_CHUNK", "VIDO_BA_SCHUR_OLD")}
        p = subprocess.run([sys.executable, "-c", code, f], env=dict(clean, **e), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        outs.append(np.load(f))
    a, b = outs
    assert list(a["it"]) == list(b["it"]) and a["it"][0] >= 2, (a["it"], b["it"])
    assert rel(a["cam"], b["cam"]) < 1e-9 and rel(a["pt"], b["pt"]) < 1e-9
    assert abs(a["chi"][1] - b["chi"][1]) <= 1e-9 * b["chi"][1] and a["chi"][1] < 0.5 * a["chi"][0]
