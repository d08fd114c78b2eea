"""CPU, world_size 2, gloo: the landmark-sharded global-BA contract.  Every rank linearises only the
landmarks of its contiguous id range (rank 0 also owns the odometry/prior factors and the lambda*I term); the summed
partial reduced systems must equal the unsharded one, and the replicated solve then gives every rank the same step.
The per-shard arithmetic is done by the CPU oracle here (there is no GPU in this container); on the GPU the same
decomposition runs in vido_ba_optimize with torch.distributed's nccl(=RCCL) backend behind the all-reduce hook."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import vido_slam_amd as V
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = V.problems.synth_ba_problem(n_cam=9, n_pt=260, kind="global", track_len=5, seed=21)
    shards = V.landmark_shards(pr["obs_pt"], pr["n_pt"], world)
    lo, hi = shards[rank]
    lam = 0.37
    S, r, chi = O.ba_reduced_system(pr, lam, lo, hi, with_cam_factors=(rank == 0))
    buf = torch.from_numpy(np.concatenate([S.ravel(), r, [chi]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    # replicated solve: identical inputs -> identical step on every rank, no broadcast needed
    n6 = 6 * pr["n_cam"]
    Sg = buf[:n6 * n6].numpy().reshape(n6, n6); rg = buf[n6 * n6:n6 * n6 + n6].numpy()
    x = np.linalg.solve(Sg, rg)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), S=Sg, r=rg, chi=buf[-1].item(), x=x, lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_rank_shards_sum_to_the_unsharded_system(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import vido_slam_amd as V
    from oracle import pyoracle as O
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz"); b = np.load(tmp_path / "rank1.npz")
    assert (a["lo"], a["hi"]) != (b["lo"], b["hi"]) and a["hi"] == b["lo"]
    assert np.array_equal(a["S"], b["S"]) and np.array_equal(a["x"], b["x"])           # replicated, bit-identical
    pr = V.problems.synth_ba_problem(n_cam=9, n_pt=260, kind="global", track_len=5, seed=21)
    S, r, chi = O.ba_reduced_system(pr, 0.37)
    assert np.allclose(a["S"], S, rtol=1e-12, atol=1e-9) and np.allclose(a["r"], r, rtol=1e-12, atol=1e-9)
    assert abs(a["chi"] - chi) < 1e-9 * max(1.0, chi)
    assert np.allclose(a["S"], a["S"].T, atol=1e-9)


def test_landmark_shards_are_contiguous_and_balanced():
    sys.path.insert(0, ROOT)
    import vido_slam_amd as V
    pr = V.problems.synth_ba_problem(n_cam=40, n_pt=4000, kind="global", track_len=10, seed=2)
    for world in (1, 2, 4, 8):
        sh = V.landmark_shards(pr["obs_pt"], pr["n_pt"], world)
        assert sh[0][0] == 0 and sh[-1][1] == pr["n_pt"] and all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        cnt = np.bincount(pr["obs_pt"], minlength=pr["n_pt"])
        loads = [cnt[lo:hi].sum() for lo, hi in sh]
        assert max(loads) <= 1.15 * (sum(loads) / world) + 50
