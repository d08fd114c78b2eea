#!/usr/bin/env python3
"""bench.py — driver contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.

Workload (BASELINE.json configs[1]): "ORB pyramid + flow-guided tracking only, synthetic 640x480
stream, 1 MI355X".  A step = one pass of the hot path over one batch of `--batch` synthetic frames
that are already resident in HBM when the timed region starts.  The per-frame path does not shard
(frame k depends on frame k-1 state): with --gpus N each rank runs an independent
replica of the same stream on its own GPU ("replicas only"), no data-path collective; value = all
frames processed by all ranks / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel, algorithmic bytes / live HIP-event
duration vs the 8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle — a scalar port, 1 core — timed on
a bounded sample of the same frames, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per step (in flight on one GPU)")
    ap.add_argument("--cpu-frames", type=int, default=200, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    import vido_slam_amd as V
    from vido_slam_amd import synth

    B, W, H = args.batch, args.width, args.height
    ctx = V.Context(device=local_rank, width=W, height=H, max_batch=B)
    # B distinct frames of a synthetic stream (rank-dependent seed: independent replicas), resident in HBM
    n_distinct = min(B, 16)
    base = synth.make_batch(n_distinct, W, H, seed=1 + 100 * rank)
    frames_host = np.ascontiguousarray(base[np.arange(B) % n_distinct])
    frames_dev = torch.from_numpy(frames_host).cuda()
    dev_arg = (frames_dev.data_ptr(), B, H, W, H * W, W)

    def step():
        return ctx.orb_extract_batch(dev_arg, want_desc=True)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage = {}
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kps, desc, cnt = step()
        for k, v in ctx.orb_timing().items():
            stage[k] = stage.get(k, 0.0) + v
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage = {k: v / max(args.steps, 1) for k, v in stage.items()}

    frames_total = B * args.steps * world
    fps = frames_total / dt
    # ---- roofline of the dominant kernel (k_fast_cells): algorithmic bytes = every pyramid pixel read
    # once (950 532 B / 640x480 frame) + 4 B per emitted candidate, per launch of B frames.
    p_px = 0
    lw, lh = W, H
    import ctypes as C
    for l in range(ctx.cfg.n_levels):
        a, b = C.c_int(), C.c_int()
        ctx.lib.vido_orb_level_size(ctx.h, l, C.byref(a), C.byref(b))
        p_px += a.value * b.value
    n_cand = stage.get("n_candidates", 0.0)
    fast_bytes = p_px * B + 4.0 * n_cand
    fast_s = stage["fast_ms"] * 1e-3
    achieved = fast_bytes / fast_s / 1e9 if fast_s > 0 else 0.0
    roofline = {"kernel": "k_fast_cells", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": int(fast_bytes), "avg_launch_ms": round(stage["fast_ms"], 4)}

    out = {
        "metric": "frames/sec end-to-end (flow+depth+track+local-BA) at 640x480; BA iters/sec",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: ORB pyramid (8 levels x1.2, 2000 features, FAST 20/7, quadtree, IC angle, 7x7 blur, rBRIEF) on a synthetic %dx%d stream; stages built so far: ORB extraction; flow-guided tracking/BA stages are added as they land" % (W, H),
                   "frames_per_step": B, "parallelism": "replicas x%d (per-frame path does not shard)" % world,
                   "inputs": "gray u8 frames resident in HBM"},
        "stage_ms_per_step": {k: round(v, 4) for k, v in stage.items() if k != "n_candidates"},
        "keypoints_per_frame": float(cnt.mean()),
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import pyoracle as O
        p = O.orb_params(n_features=ctx.cfg.n_features, scale_factor=ctx.cfg.scale_factor, n_levels=ctx.cfg.n_levels,
                         ini_th=ctx.cfg.ini_th_fast, min_th=ctx.cfg.min_th_fast)
        t1 = time.perf_counter()
        for i in range(args.cpu_frames):
            O.orb_extract(p, frames_host[i % B])
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(args.cpu_frames / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d of the same 640x480 frames through oracle/orb_oracle.c (scalar C restatement of ORBextractor::operator(), descriptors on), %.1f s" % (args.cpu_frames, cdt)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
