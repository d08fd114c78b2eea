#!/usr/bin/env python3
"""Golden vectors for row N3: runs the REFERENCE maskrcnn_benchmark detector (imported from /root/reference
in this container only) on a small seeded image with deterministic weights, and stores stage-wise outputs in
tests/golden/maskrcnn_graph.npz.  Nothing of the reference travels: the fixture is data.

The reference needs modules this image lacks; they are replaced as follows (none of them carries model arithmetic):
  yacs.config.CfgNode   -> a 40-line attribute dict with merge_from_file/merge_from_list (config plumbing only)
  apex.amp, cv2, memory_profiler -> empty stubs (decorators / unused imports on this path)
  maskrcnn_benchmark._C -> nms and roi_align_forward are the C oracle restatements (oracle/nets_oracle.c), themselves pinned
                           by the reference's own known-answer tests (tests/golden/maskrcnn_kats.npz)
The graph is built from the node's own config (src/configs/caffe2/e2e_mask_rcnn_X_101_32x8d_FPN_1x_caffe2.yaml) shrunk
through the reference's config keys (R-50-FPN depth, 4 groups x 4, 16 FPN channels, 7 classes, 20 detections) so that the
fixture stays small; the full-size graph is checked for state-dict identity (567 entries, 107 837 937 elements)."""
import copy, os, sys, types
import numpy as np
np.float = float                                   # the reference predates numpy 1.24
import torch, yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/thirdparty/mask_rcnn"
sys.path.insert(0, REF); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import pyoracle                                    # noqa: E402
import vido_slam_amd                               # noqa: E402
from vido_slam_amd.nets.weights import fill_maskrcnn   # noqa: E402
from vido_slam_amd.synth import make_canvas        # noqa: E402

TINY = ["MODEL.DEVICE", "cpu", "MODEL.BACKBONE.CONV_BODY", "R-50-FPN", "MODEL.RESNETS.NUM_GROUPS", 4, "MODEL.RESNETS.WIDTH_PER_GROUP", 4,
        "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
        "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 64, "MODEL.ROI_BOX_HEAD.NUM_CLASSES", 7, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16, 16, 16),
        "MODEL.ROI_HEADS.DETECTIONS_PER_IMG", 20]


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v
    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)
    def __setattr__(self, k, v):
        self[k] = v
    def clone(self):
        return copy.deepcopy(self)
    def freeze(self):
        pass
    def merge_from_file(self, path):
        def lit(v):
            if isinstance(v, str):
                try:
                    return eval(v, {}, {})
                except Exception:
                    return v
            return v
        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    merge(dst[k], v)
                else:
                    dst[k] = lit(v)
        merge(self, yaml.safe_load(open(path)))
    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            d = self; ks = k.split(".")
            for kk in ks[:-1]:
                d = d[kk]
            d[ks[-1]] = v


def install_stubs():
    yacs = types.ModuleType("yacs"); yc = types.ModuleType("yacs.config"); yc.CfgNode = CfgNode; yacs.config = yc
    apex = types.ModuleType("apex"); amp = types.ModuleType("apex.amp"); amp.float_function = lambda f: f; apex.amp = amp
    mp = types.ModuleType("memory_profiler"); mp.profile = lambda f=None, **k: (f if f is not None else (lambda g: g))
    sys.modules.update({"yacs": yacs, "yacs.config": yc, "apex": apex, "apex.amp": amp, "cv2": types.ModuleType("cv2"), "memory_profiler": mp})
    _C = types.ModuleType("maskrcnn_benchmark._C")
    _C.nms = lambda boxes, scores, thresh: torch.from_numpy(pyoracle.nms(boxes.numpy(), scores.numpy(), float(thresh)).astype(np.int64))
    _C.roi_align_forward = lambda inp, rois, scale, ph, pw, sr: torch.from_numpy(pyoracle.roi_align(inp.numpy(), rois.numpy(), float(scale), int(ph), int(pw), int(sr)))
    import maskrcnn_benchmark
    sys.modules["maskrcnn_benchmark._C"] = _C; maskrcnn_benchmark._C = _C


def build(overrides):
    from maskrcnn_benchmark.config import cfg as base
    cfg = base.clone()
    cfg.merge_from_file(REF + "/src/configs/caffe2/e2e_mask_rcnn_X_101_32x8d_FPN_1x_caffe2.yaml")
    cfg.merge_from_list(overrides)
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    return build_detection_model(cfg).eval()


def main():
    torch.set_grad_enabled(False)
    install_stubs()
    full = build(["MODEL.DEVICE", "cpu"])
    sd = full.state_dict()
    out = dict(full_keys=np.array(list(sd.keys())), full_shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
               full_cell_anchors=np.stack([sd["rpn.anchor_generator.cell_anchors.%d" % i].numpy() for i in range(5)]))
    del full, sd
    model = build(TINY)
    fill_maskrcnn(model, seed=31)
    H, W = 96, 128
    c = make_canvas(H + 6, W + 6, seed=5, n_rect=25).astype(np.float32)
    img = np.stack([c[:H, :W], c[3:H + 3, 3:W + 3], c[6:, 6:]], 0) / np.float32(255.0)      # O(1) activations under random weights
    t = torch.from_numpy(img)[None]
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.inference import Masker
    images = to_image_list(t, 0)
    feats = model.backbone(images.tensors)
    proposals = model.rpn(images, feats)[0]
    x, dets, _ = model.roi_heads.box(feats, proposals)
    _, result, _ = model.roi_heads.mask(feats, dets)
    r = result[0]
    print("levels", [tuple(f.shape) for f in feats], "proposals", len(proposals[0]), "detections", len(r), "labels", sorted(set(r.get_field("labels").tolist())))
    out.update(image=img, seed=np.int32(31))
    for i, f in enumerate(feats):
        out["feat%d" % i] = f.numpy()
    out.update(proposals=proposals[0].bbox.numpy(), objectness=proposals[0].get_field("objectness").numpy(),
               det_boxes=r.bbox.numpy(), det_scores=r.get_field("scores").numpy(), det_labels=r.get_field("labels").numpy(),
               det_masks=r.get_field("mask").numpy())
    # predictor.py:246-252 + run_mask_rcnn.py:93-118: resize to the "original" frame, paste, label image (all detections kept)
    OW, OH = 200, 120
    rr = r.resize((OW, OH))
    pasted = Masker(threshold=0.5, padding=1)([rr.get_field("mask")], [rr])[0]
    label = np.zeros((OH, OW), np.uint8)
    for m, l in zip(pasted, rr.get_field("labels").numpy()):
        label += m[0].astype(np.uint8) * np.uint8(l)
    out.update(paste_size=np.array([OW, OH]), resized_boxes=rr.bbox.numpy(), pasted=np.packbits(np.asarray(pasted)[:, 0].astype(bool), axis=-1), label_image=label)
    print("label image histogram", np.bincount(label.reshape(-1))[:12])
    np.savez_compressed(os.path.join(REPO, "tests/golden/maskrcnn_graph.npz"), **out)
    print("wrote tests/golden/maskrcnn_graph.npz", os.path.getsize(os.path.join(REPO, "tests/golden/maskrcnn_graph.npz")))


if __name__ == "__main__":
    main()
