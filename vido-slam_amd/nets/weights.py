"""Deterministic random-init weights (no checkpoints ship with the reference and there is no network): every tensor
of a state dict is drawn from a numpy RandomState seeded by crc32(key), so the fixture generator (which fills the
REFERENCE modules) and the tests/bench (which fill ours) get identical values from the key names alone."""
import zlib
import numpy as np
import torch


def deterministic_tensor(key, shape, seed=0):
    rng = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        return torch.from_numpy(rng.uniform(0.8, 1.2, shape).astype(np.float32))
    if leaf == "running_mean":
        return torch.from_numpy(rng.uniform(-0.1, 0.1, shape).astype(np.float32))
    if len(shape) == 1:                                    # bias / BatchNorm affine
        if leaf == "weight":
            return torch.from_numpy(rng.uniform(0.8, 1.2, shape).astype(np.float32))
        return torch.from_numpy(rng.uniform(-0.05, 0.05, shape).astype(np.float32))
    fan_in = int(np.prod(shape[1:]))
    b = np.sqrt(6.0 / fan_in)                           # He-uniform keeps activations O(1) through the LeakyReLU/ELU stacks
    return torch.from_numpy(rng.uniform(-b, b, shape).astype(np.float32))


def fill_deterministic(module, seed=0, keep=("cell_anchors",)):
    """keys containing one of `keep` are left as constructed (Mask R-CNN's anchor tables are buffers, not weights)"""
    state = {k: (v if any(s in k for s in keep) else deterministic_tensor(k, v.shape, seed)) for k, v in module.state_dict().items()}
    module.load_state_dict(state)
    return module


def fill_maskrcnn(module, seed=0, reg_scale=0.05):
    """fill_deterministic + small box-regression / logit layers: with He-uniform heads every proposal collapses onto the
    image border and objectness / class scores saturate to exactly 1.0 (ties, whose order torch.topk leaves to the backend).
    Scaling the bbox_pred, cls_logits and cls_score layers keeps proposals near their anchors and scores distinct, so NMS /
    level mapping / mask pasting see varied, well-ordered boxes (used identically by the fixture generator, tests, bench)."""
    fill_deterministic(module, seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if "bbox_pred" in k or "cls_logits" in k or "cls_score" in k:
                v.mul_(reg_scale)
            elif k.endswith("bn3.weight"):
                v.mul_(0.25)                # damped residual branches: activations stay O(1) through 16..33 bottlenecks
    return module
