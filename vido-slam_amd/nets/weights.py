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


@torch.no_grad()
def calibrate_detector_scores(net, image, target_std=2.0):
    """Rescale the box head's class-logit layer so that the logits of `image`'s proposals have standard deviation `target_std`.
    With the plain deterministic fill the activations grow through the un-normalised FPN / two-FC head and every class score saturates to exactly 1.0 in fp32; all
    detections then TIE at the kthvalue cut of box_head/inference.py:131-137 and the detections_per_img = 100 cap does not bind (200-300 detections per frame, a workload
    trained weights cannot produce).  With distinct scores the cap binds: the mask head sees at most 100 detections per image, as in the reference.  Only cls_score is
    touched (one scalar); the state-dict layout is unchanged.  Returns the factor applied."""
    feats, logits, deltas = net.trunk(image)
    H, W = image.shape[-2:]
    proposals, objectness = net.rpn.proposals(feats, logits, deltas, (W, H))
    box = net.roi_heads.box
    cls_logits, _ = box.predictor(box.feature_extractor(feats[:len(net.config.pool_scales)], proposals))
    valid = objectness >= 0 if objectness is not None else torch.ones(len(proposals), dtype=torch.bool, device=cls_logits.device)
    std = float(cls_logits[valid].std())
    k = target_std / std if std > 0 else 1.0
    box.predictor.cls_score.weight.mul_(k); box.predictor.cls_score.bias.mul_(k)
    return k
