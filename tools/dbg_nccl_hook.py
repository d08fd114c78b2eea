"""world_size=1 RCCL smoke of the all-reduce hook used by the sharded global BA (zero-copy wrap of a raw device pointer)."""
import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import vido_slam_amd as V
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = V.Context()
pr = V.problems.synth_ba_problem(n_cam=30, n_pt=1500, kind="global", track_len=8, seed=4)
a = V.ba_optimize(ctx, pr)
b = V.ba_optimize(ctx, pr, rank=0, world=1, shard=(0, pr["n_pt"]), allreduce=V.torch_allreduce_hook())
print("iters", a["iterations"], b["iterations"], "max diff", np.abs(a["cam_T"] - b["cam_T"]).max(), "chi2", a["chi2_final"], b["chi2_final"])
# pointer wrap sanity
t = torch.arange(8, dtype=torch.float64, device="cuda")
w = torch.as_tensor(V.host._DevBuf(t.data_ptr(), 8), device="cuda")
w += 1
print("zero-copy wrap ok:", bool((t == torch.arange(1, 9, device="cuda")).all()))
dist.destroy_process_group()
