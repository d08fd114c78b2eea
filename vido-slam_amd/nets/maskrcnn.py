"""Mask R-CNN inference graph (reference src/thirdparty/mask_rcnn: maskrcnn_benchmark/modeling/** as configured by
src/configs/caffe2/e2e_mask_rcnn_X_101_32x8d_FPN_1x_caffe2.yaml + config/defaults.py, driven by src/predictor.py:215-262 and
src/run_mask_rcnn.py:76-127).  Forward only, tensors instead of BoxList objects, one image per call (the node processes one
frame at a time).  Module/parameter names equal the reference's (backbone.body.layer3.22.conv2.weight,
rpn.anchor_generator.cell_anchors.0, roi_heads.mask.predictor.conv5_mask.bias ...), so its checkpoints load unchanged.
ROI-Align, NMS and box decoding go through `ops` (HipOps: the HIP kernels of libvido_slam_hip.so); there is no CPU path
in the product — the CPU tests inject oracle-backed ops."""
import math
import os
from collections import OrderedDict
from dataclasses import dataclass
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class MaskRCNNConfig:
    # backbone (modeling/backbone/resnet.py:64-130, fpn.py)
    blocks: tuple = (3, 4, 23, 3)            # R-101-FPN
    groups: int = 32
    width_per_group: int = 8
    stride_in_1x1: bool = False
    stem_out: int = 64
    res2_out: int = 256
    fpn_out: int = 256
    # RPN (config/defaults.py:128-175, rpn/*.py)
    anchor_sizes: tuple = (32, 64, 128, 256, 512)
    aspect_ratios: tuple = (0.5, 1.0, 2.0)
    anchor_strides: tuple = (4, 8, 16, 32, 64)
    pre_nms_top_n: int = 1000
    post_nms_top_n: int = 1000
    fpn_post_nms_top_n: int = 1000
    rpn_nms: float = 0.7
    rpn_min_size: int = 0
    # box head (defaults.py:198-232)
    pool_scales: tuple = (0.25, 0.125, 0.0625, 0.03125)
    box_resolution: int = 7
    sampling_ratio: int = 2
    mlp_dim: int = 1024
    num_classes: int = 81
    score_thresh: float = 0.05
    nms: float = 0.5
    detections_per_img: int = 100
    bbox_reg_weights: tuple = (10.0, 10.0, 5.0, 5.0)
    # mask head (defaults.py:234-256)
    mask_resolution: int = 14
    mask_layers: tuple = (256, 256, 256, 256)


class FrozenBatchNorm2d(nn.Module):        # layers/batch_norm.py:6-31
    def __init__(self, n):
        super().__init__()
        for name, v in (("weight", torch.ones(n)), ("bias", torch.zeros(n)), ("running_mean", torch.zeros(n)), ("running_var", torch.ones(n))):
            self.register_buffer(name, v)

    def forward(self, x):
        scale = self.weight * self.running_var.rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


class _Bottleneck(nn.Module):              # resnet.py:277-372 (BottleneckWithFixedBatchNorm)
    def __init__(self, cin, mid, cout, groups, stride_in_1x1, stride):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = nn.Conv2d(cin, mid, 1, s1, bias=False); self.bn1 = FrozenBatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, s3, 1, bias=False, groups=groups); self.bn2 = FrozenBatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = FrozenBatchNorm2d(cout)

    _ep = None                             # nets/fuse.py::fold_batchnorm installs the folded tensors + the fused HIP epilogue
    _w2p = None                            # ... and conv2's weight in the operand order of csrc/gconv.hip where that kernel takes the layer
    _w1p = _w3p = _wdp = None; _c1x1_min_tiles = 0      # ... and the 1x1 convolutions' in the operand order of csrc/conv1x1.hip (fuse.py decides which layers take it)

    def forward(self, x):
        if self._ep is not None and x.is_cuda:
            ep = self._ep; c1, c2 = self.conv1, self.conv2; ops = self._ops
            b1x = x.shape[0] == 1
            hw_in = x.shape[2] * x.shape[3]
            b1_done = False
            from .ops import conv1x1_fills_chip as fills
            if self._w1p is not None and b1x and ops.conv1x1_supported(x.shape[1], self._w1.shape[0], hw_in) and fills(self._w1.shape[0], hw_in):
                y = ops.conv1x1_bias_act(x.contiguous(), self._w1p, self._b1, None, 0.0); b1_done = True      # our GEMM: its bias + ReLU leave through its accumulators
            else:
                y = F.conv2d(x, self._w1, None, c1.stride)                                                   # the library's: they ride on conv2's operand reads, or run below
            if (self._w2p is not None and tuple(c2.stride) == (2, 2)):
                if b1_done and y.shape[0] == 1 and self._ops.gconv3x3_s2_supported(y.shape[2], y.shape[3], c2.in_channels // c2.groups, c2.out_channels // c2.groups):
                    y = self._ops.gconv3x3_s2_bias_act(y, self._w2p, self._b2, c2.groups, 0.0)
                else:
                    y = ep(F.conv2d(y if b1_done else ep(y, self._b1, None, 0.0), self._w2, None, c2.stride, c2.padding, c2.dilation, c2.groups), self._b2, None, 0.0)
            elif self._w2p is not None and y.shape[0] == 1 and self._ops.gconv3x3_supported(y.shape[2], y.shape[3], c2.in_channels // c2.groups, c2.out_channels // c2.groups):
                # conv2's own bias + ReLU leave through its accumulators; after a library conv1 its bias + ReLU are applied where conv2 reads its operands
                y = self._ops.gconv3x3_bias_act(y, self._w2p, self._b2, c2.groups, 0.0, in_bias=None if b1_done else self._b1)
            else:
                y = ep(F.conv2d(y if b1_done else ep(y, self._b1, None, 0.0), self._w2, None, c2.stride, c2.padding, c2.dilation, c2.groups), self._b2, None, 0.0)
            if self.downsample is None:
                sc = x
            elif self._wdp is not None and b1x and tuple(self.downsample[0].stride) == (1, 1) and ops.conv1x1_supported(x.shape[1], self._wd.shape[0], hw_in) and fills(self._wd.shape[0], hw_in):
                sc = ops.conv1x1_bias_act(x.contiguous(), self._wdp, self._bd, None, 1.0)
            elif (self._wdp is not None and b1x and tuple(self.downsample[0].stride) == (2, 2) and ops.conv1x1_supported(x.shape[1], self._wd.shape[0], ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2))
                  and fills(self._wd.shape[0], ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2))):
                # a 1x1 convolution with stride 2 reads every other row and column: one strided copy (a quarter of the map), then the GEMM with the bias in its epilogue — the
                # library's strided form runs at 36 TFLOP/s (100 us for 3.6 GFLOP) and needs the bias pass behind it
                sc = ops.conv1x1_bias_act(x[:, :, ::2, ::2].contiguous(), self._wdp, self._bd, None, 1.0)
            else:
                sc = ep(F.conv2d(x, self._wd, None, self.downsample[0].stride), self._bd, None, 1.0)
            if (self._w3p is not None and y.shape[0] == 1 and ops.conv1x1_supported(y.shape[1], self._w3.shape[0], y.shape[2] * y.shape[3])
                    and (fills(self._w3.shape[0], y.shape[2] * y.shape[3]) or self._c1x1_min_tiles < 0)):
                return ops.conv1x1_bias_act(y.contiguous(), self._w3p, self._b3, sc, 0.0)      # bias + shortcut + ReLU leave through the GEMM's accumulators: no pass over the output
            return ep(F.conv2d(y, self._w3), self._b3, sc.contiguous(), 0.0)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _Stem(nn.Module):                    # resnet.py:375-395
    def __init__(self, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(3, cout, 7, 2, 3, bias=False); self.bn1 = FrozenBatchNorm2d(cout)

    _ep = None

    def forward(self, x):
        if self._ep is not None and x.is_cuda:
            return F.max_pool2d(self._ep(F.conv2d(x, self._w1, None, 2, 3), self._b1, None, 0.0), 3, 2, 1)
        return F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)


class _Body(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.stem = _Stem(c.stem_out)
        cin = c.stem_out
        for i, n in enumerate(c.blocks, 1):
            f = 2 ** (i - 1)
            mid, cout = c.groups * c.width_per_group * f, c.res2_out * f
            blocks = []
            for b in range(n):
                blocks.append(_Bottleneck(cin, mid, cout, c.groups, c.stride_in_1x1, (2 if i > 1 else 1) if b == 0 else 1))
                cin = cout
            setattr(self, "layer%d" % i, nn.Sequential(*blocks))
        self.n_stages = len(c.blocks)

    def forward(self, x):
        x = self.stem(x); out = []
        for i in range(1, self.n_stages + 1):
            x = getattr(self, "layer%d" % i)(x); out.append(x)
        return out


class _FPN(nn.Module):                     # fpn.py:7-82 with LastLevelMaxPool
    def __init__(self, c):
        super().__init__()
        self.n = len(c.blocks)
        for i in range(1, self.n + 1):
            setattr(self, "fpn_inner%d" % i, nn.Conv2d(c.res2_out * 2 ** (i - 1), c.fpn_out, 1))
            setattr(self, "fpn_layer%d" % i, nn.Conv2d(c.fpn_out, c.fpn_out, 3, 1, 1))

    _ops = None                            # MaskRCNN.__init__: HipOps — the output convolutions as Winograd launches on the matrix pipe (csrc/wino.hip)

    def _layer(self, i, x):
        conv = getattr(self, "fpn_layer%d" % i)
        if self._ops is not None and x.is_cuda and hasattr(self._ops, "wino3x3_conv") and not os.environ.get("VIDO_NO_WINO"):
            y = self._ops.wino3x3_conv(conv, x, 1.0)
            if y is not None:
                return y
        return conv(x)

    def _inner(self, i, x, top=None):
        """fpn.py:55-66: the lateral 1x1 convolution (+ bias) + the nearest-upsampled coarser level — one GEMM launch with both in its epilogue (csrc/conv1x1.hip) where the
        maps have that form (even sizes), else the library convolution, the upsampling and the sum."""
        conv = getattr(self, "fpn_inner%d" % i)
        if self._ops is not None and x.is_cuda and hasattr(self._ops, "conv1x1_conv") and not os.environ.get("VIDO_NO_CONV1X1"):
            if top is None:
                y = self._ops.conv1x1_conv(conv, x, 1.0)
            elif x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and tuple(top.shape[2:]) == (x.shape[2] // 2, x.shape[3] // 2):
                y = self._ops.conv1x1_conv(conv, x, 1.0, residual_up2=top)
            else:
                y = None
            if y is not None:
                return y
        y = conv(x)
        return y if top is None else y + F.interpolate(top, scale_factor=2, mode="nearest")

    def forward(self, feats):
        inner = self._inner(self.n, feats[-1])
        out = [self._layer(self.n, inner)]
        for i in range(self.n - 1, 0, -1):
            inner = self._inner(i, feats[i - 1], inner)
            out.insert(0, self._layer(i, inner))
        out.append(F.max_pool2d(out[-1], 1, 2, 0))
        return out


def cell_anchors(stride, size, ratios):
    """rpn/anchor_generator.py:219-292 (Detectron's generate_anchors): one scale per FPN level, float64 then .float()."""
    base = np.array([0.0, 0.0, stride - 1.0, stride - 1.0])
    w, h = base[2] - base[0] + 1, base[3] - base[1] + 1
    xc, yc = base[0] + 0.5 * (w - 1), base[1] + 0.5 * (h - 1)
    ws = np.round(np.sqrt(w * h / np.asarray(ratios, np.float64))); hs = np.round(ws * np.asarray(ratios, np.float64))
    s = float(size) / stride
    rows = [[xc - 0.5 * (a * s - 1), yc - 0.5 * (b * s - 1), xc + 0.5 * (a * s - 1), yc + 0.5 * (b * s - 1)] for a, b in zip(ws, hs)]
    return torch.from_numpy(np.asarray(rows, np.float64)).float()


class _Buffers(nn.Module):
    def __init__(self, tensors):
        super().__init__()
        for i, t in enumerate(tensors):
            self.register_buffer(str(i), t)

    def __iter__(self):
        return iter(self._buffers.values())


class _AnchorGenerator(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.strides = c.anchor_strides
        self.cell_anchors = _Buffers([cell_anchors(st, sz, c.aspect_ratios) for st, sz in zip(c.anchor_strides, c.anchor_sizes)])

    def forward(self, grid_sizes):          # anchor_generator.py:77-101; order (y, x, anchor)
        out = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=base.device)
            sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=base.device)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), 1)
            out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
        return out


class _RPNHead(nn.Module):                 # rpn/rpn.py:74-107
    def __init__(self, ch, na):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, 1, 1); self.cls_logits = nn.Conv2d(ch, na, 1); self.bbox_pred = nn.Conv2d(ch, na * 4, 1)

    _ops = None                            # MaskRCNN.__init__: HipOps, for the one-pass bias + ReLU

    def forward(self, feats):
        t = [_conv_bias_relu(self.conv, f, self._ops) for f in feats]
        ops = self._ops
        if ops is not None and t[0].is_cuda and hasattr(ops, "conv1x1_skinny") and not os.environ.get("VIDO_NO_CONVSMALL") and t[0].shape[0] == 1 and self.conv.out_channels % 2 == 0 and self.conv.out_channels <= 256:
            # objectness + box deltas (3 + 12 output channels) as ONE few-output-channel 1x1 launch per level that reads the head's feature map once (csrc/convsmall.hip); the two
            # results are the leading / trailing planes of its output.  The library runs two convolutions of 36 + 46 us at P2 for 56 MB of input.
            from .ops import pack_conv1x1_skinny
            key = (self.cls_logits.weight.data_ptr(), self.cls_logits.weight._version, self.bbox_pred.weight.data_ptr(), self.bbox_pred.weight._version, str(t[0].device))
            if getattr(self, "_head_key", None) != key:
                self._head_w = pack_conv1x1_skinny(torch.cat([self.cls_logits.weight, self.bbox_pred.weight], 0)).to(t[0].device)
                self._head_b = torch.cat([self.cls_logits.bias, self.bbox_pred.bias], 0).detach().contiguous(); self._head_key = key
            na = self.cls_logits.out_channels; nt = na + self.bbox_pred.out_channels
            ys = [ops.conv1x1_skinny(x, self._head_w, self._head_b, nt, 1.0) for x in t]
            return [y[:, :na] for y in ys], [y[:, na:] for y in ys]
        return [self.cls_logits(x) for x in t], [self.bbox_pred(x) for x in t]


def _conv_bias_relu(conv, x, ops):
    """relu(conv(x)); on the device the library convolution runs without its bias and the bias + ReLU are ONE in-place pass (vido_bias_act) instead of an add and a clamp."""
    if ops is None or not x.is_cuda or conv.bias is None:
        return F.relu(conv(x))
    if isinstance(conv, nn.Conv2d) and hasattr(ops, "wino3x3_conv") and not os.environ.get("VIDO_NO_WINO"):
        y = ops.wino3x3_conv(conv, x, 0.0)              # dense 3x3: Winograd on the matrix pipe with bias + ReLU in its epilogue (csrc/wino.hip)
        if y is not None:
            return y
    if isinstance(conv, nn.ConvTranspose2d) and hasattr(ops, "deconv2x2_conv"):
        y = ops.deconv2x2_conv(conv, x, 0.0)           # the mask head's 2 x 2 stride-2 transposed convolution: one split-fp16 GEMM with a scatter epilogue (csrc/conv1x1.hip)
        if y is not None:
            return y
    if isinstance(conv, nn.ConvTranspose2d):
        y = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding, conv.groups, conv.dilation)
    else:
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return ops.bias_act_(y.contiguous(), conv.bias, 0.0)


def clip_boxes(b, w, h):                    # structures/bounding_box.py:214-224 (TO_REMOVE = 1)
    b = b.clone()
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=w - 1); b[:, 1::2] = b[:, 1::2].clamp(min=0, max=h - 1)
    return b


def _topk_stable(v, k):
    """topk(sorted=True) with ties broken by the lower index — torch.topk leaves the tie order to the backend (CPU and
    GPU differ), and saturated objectness (sigmoid == 1.0) makes ties common."""
    s, idx = torch.sort(v, descending=True, stable=True)
    return s[:k], idx[:k]


class _RPN(nn.Module):
    def __init__(self, c, ops):
        super().__init__()
        self.c, self.ops = c, ops
        self.anchor_generator = _AnchorGenerator(c)
        self.head = _RPNHead(c.fpn_out, len(c.aspect_ratios))

    def forward(self, feats, image_wh):     # rpn/inference.py:73-159 (test path, one image)
        logits, deltas = self.head(feats)
        return self.proposals(feats, logits, deltas, image_wh)

    def proposals(self, feats, logits, deltas, image_wh):
        if hasattr(self.ops, "nms_segments") and logits[0].is_cuda and self.c.rpn_min_size <= 0:
            return self.proposals_device(feats, logits, deltas, image_wh)
        c = self.c; W, H = image_wh
        anchors = self.anchor_generator([f.shape[-2:] for f in feats])
        boxes, scores = [], []
        for a, lo, de in zip(anchors, logits, deltas):
            A, h, w = lo.shape[1:]
            obj = lo[0].permute(1, 2, 0).reshape(-1).sigmoid()                       # (y, x, anchor)
            reg = de[0].view(A, 4, h, w).permute(2, 3, 0, 1).reshape(-1, 4)
            k = min(c.pre_nms_top_n, obj.numel())
            obj, idx = _topk_stable(obj, k)
            prop = clip_boxes(self.ops.box_decode(reg[idx], a[idx], (1.0, 1.0, 1.0, 1.0)), W, H)
            keep = ((prop[:, 2] - prop[:, 0] + 1 >= c.rpn_min_size) & (prop[:, 3] - prop[:, 1] + 1 >= c.rpn_min_size)).nonzero().squeeze(1)
            prop, obj = prop[keep], obj[keep]
            keep = self.ops.nms(prop, obj, c.rpn_nms)[: c.post_nms_top_n]
            boxes.append(prop[keep]); scores.append(obj[keep])
        boxes, scores = torch.cat(boxes), torch.cat(scores)
        k = min(c.fpn_post_nms_top_n, scores.numel())                                 # select_over_all_levels, test branch
        _, idx = _topk_stable(scores, k)
        return boxes[idx], scores[idx]


def _rpn_proposals_device(self, feats, logits, deltas, image_wh):
    """proposals() without a host round trip: the five levels are decoded into one [levels * k] batch, ONE segmented NMS replaces the five layers.nms
    calls (each of which ended in a device-to-host count), suppressed entries are carried with objectness -1 instead of being compacted, and the final
    top-k over all levels sorts them to the end.  Same proposals in the same order as the reference's loop (rpn/inference.py:73-159) whenever at least
    fpn_post_nms_top_n boxes survive in total (the count is returned on the device; fewer survivors leave trailing zero-area boxes with objectness -1,
    which the box head masks out)."""
    c = self.c; W, H = image_wh
    anchors = self.anchor_generator([f.shape[-2:] for f in feats])
    K = c.pre_nms_top_n
    props, objs, ns = [], [], []
    for a, lo, de in zip(anchors, logits, deltas):
        A, h, w = lo.shape[1:]
        obj = lo[0].permute(1, 2, 0).reshape(-1).sigmoid()
        reg = de[0].view(A, 4, h, w).permute(2, 3, 0, 1).reshape(-1, 4)
        k = min(K, obj.numel())
        obj, idx = _topk_stable(obj, k)
        prop = clip_boxes(self.ops.box_decode(reg[idx], a[idx], (1.0, 1.0, 1.0, 1.0)), W, H)
        if k < K:                                                        # coarse levels have fewer anchors than pre_nms_top_n: pad the segment (never read by the NMS)
            prop = torch.cat([prop, prop.new_zeros((K - k, 4))]); obj = torch.cat([obj, obj.new_full((K - k,), -1.0)])
        props.append(prop); objs.append(obj); ns.append(k)
    L = len(props)
    boxes = torch.cat(props); scores = torch.cat(objs)
    dev = boxes.device
    key = (L, K, tuple(ns), str(dev))
    if getattr(self, "_seg_key", None) != key:
        self._seg_off = torch.arange(L, device=dev, dtype=torch.int32) * K; self._seg_n = torch.tensor(ns, device=dev, dtype=torch.int32); self._seg_key = key
        self._pos = torch.arange(K, device=dev, dtype=torch.int32).unsqueeze(0).expand(L, K)
    keep, cnt = self.ops.nms_segments(boxes, self._seg_off, self._seg_n, K, c.rpn_nms)
    # kept[l, p] <=> position p of level l survived: keep rows are ascending positions padded with -1
    kept = torch.zeros((L, K + 1), dtype=torch.bool, device=dev)
    kept.scatter_(1, torch.where(keep >= 0, keep, torch.full_like(keep, K)).long(), True)
    kept = kept[:, :K]
    if c.post_nms_top_n < K:                                             # [:post_nms_top_n] of every level's kept list
        kept &= (kept.cumsum(1) <= c.post_nms_top_n)
    scores = torch.where(kept.reshape(-1), scores, scores.new_full((), -1.0))
    k = min(c.fpn_post_nms_top_n, scores.numel())
    top, idx = _topk_stable(scores, k)
    valid = top >= 0
    return torch.where(valid.unsqueeze(1), boxes[idx], boxes.new_zeros(())), top


_RPN.proposals_device = _rpn_proposals_device


def _rpn_proposals_fused(self, feats, logits, deltas, image_wh):
    """proposals_device() with the selection logic in three ordered device kernels (csrc/detpost.hip) instead of ~80 torch launches: vido_rpn_select (sigmoid + top-k +
    decode + clip of all levels), the segmented NMS, vido_rpn_merge (the best fpn_post_nms_top_n over the levels).  Same rows as proposals_device()."""
    c = self.c; W, H = image_wh; K = c.pre_nms_top_n; L = len(logits); dev = logits[0].device
    boxes, scores, n = self.ops.rpn_select(logits, deltas, list(self.anchor_generator.cell_anchors), self.anchor_generator.strides, K, (W, H))
    if getattr(self, "_fseg_key", None) != (L, K, str(dev)):
        self._fseg_off = torch.arange(L, device=dev, dtype=torch.int32) * K; self._fseg_key = (L, K, str(dev))
    keep, cnt = self.ops.nms_segments(boxes, self._fseg_off, n, K, c.rpn_nms)
    props, obj, _ = self.ops.rpn_merge(boxes, scores, keep, cnt, K, c.post_nms_top_n, min(c.fpn_post_nms_top_n, L * K))
    return props, obj


_RPN.proposals_fused = _rpn_proposals_fused


class FpnMaps(tuple):
    """The four pooled FPN maps [1,C,H,W] plus their channels-last copies (.nhwc: [1,H,W,C], HipOps.to_nhwc) — what the ROI-Align kernel reads."""
    nhwc = ()


class _Pooler(nn.Module):                  # modeling/poolers.py:11-121
    def __init__(self, resolution, scales, sampling_ratio, ops):
        super().__init__()
        self.res, self.scales, self.sr, self.ops = resolution, scales, sampling_ratio, ops
        self.k_min = -math.log2(scales[0]); self.k_max = -math.log2(scales[-1])

    def forward(self, feats, boxes):
        if boxes.is_cuda and boxes.dtype == torch.float32 and hasattr(self.ops, "roi_levels") and not os.environ.get("VIDO_NO_ROI_LEVELS"):
            lvl = self.ops.roi_levels(boxes, self.k_min, self.k_max)                         # one launch instead of fourteen element-wise ones (same fp32 operations)
        else:
            area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
            lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-6)).clamp(min=self.k_min, max=self.k_max).to(torch.int64) - int(self.k_min)
        if isinstance(feats, FpnMaps):                                                      # channels-last copies made once per frame (MaskRCNN.fpn_maps): lanes across channels
            return self.ops.roi_align_fpn_nhwc(feats.nhwc, boxes, lvl, (self.res, self.res), self.scales, self.sr)
        if hasattr(self.ops, "roi_align_fpn") and boxes.is_cuda and len(feats) == 4:       # one launch, no per-level nonzero / gather / scatter
            return self.ops.roi_align_fpn(feats, boxes, lvl, (self.res, self.res), self.scales, self.sr)
        rois = torch.cat([boxes.new_zeros((len(boxes), 1)), boxes], 1)
        out = feats[0].new_zeros((len(boxes), feats[0].shape[1], self.res, self.res))
        for l, (f, s) in enumerate(zip(feats, self.scales)):
            idx = torch.nonzero(lvl == l).squeeze(1)
            if idx.numel():
                out[idx] = self.ops.roi_align(f, rois[idx], (self.res, self.res), s, self.sr)
        return out


class _BoxFeatures(nn.Module):             # roi_box_feature_extractors.py:50-81
    def __init__(self, c, ops):
        super().__init__()
        self.pooler = _Pooler(c.box_resolution, c.pool_scales, c.sampling_ratio, ops)
        self.fc6 = nn.Linear(c.fpn_out * c.box_resolution ** 2, c.mlp_dim); self.fc7 = nn.Linear(c.mlp_dim, c.mlp_dim)

    def forward(self, feats, boxes):
        x = self.pooler(feats, boxes).flatten(1)
        ops = self.pooler.ops
        # fc6 (12544 -> 1024 over the proposals: the largest library launch of the detector, 217 us) and fc7 (1024 -> 1024, 32 us) as split-fp16 GEMMs split over K
        # (csrc/fch.hip: 96 / 23 us); below 128 rows (the dynamic head on a few detections) the library's launch is as fast
        use = ops is not None and hasattr(ops, "fc_h_linear") and x.shape[0] >= 128
        y = ops.fc_h_linear(self.fc6, x, 0.0) if use else None
        if y is None:
            y = F.relu(self.fc6(x))
        z = ops.fc_h_linear(self.fc7, y, 0.0) if use else None
        return z if z is not None else F.relu(self.fc7(y))


class _BoxPredictor(nn.Module):            # roi_box_predictors.py:35-57
    def __init__(self, c):
        super().__init__()
        self.cls_score = nn.Linear(c.mlp_dim, c.num_classes); self.bbox_pred = nn.Linear(c.mlp_dim, c.num_classes * 4)

    def forward(self, x):
        return self.cls_score(x), self.bbox_pred(x)


class _BoxHead(nn.Module):
    def __init__(self, c, ops):
        super().__init__()
        self.c, self.ops = c, ops
        self.feature_extractor = _BoxFeatures(c, ops); self.predictor = _BoxPredictor(c)

    def forward(self, feats, proposals, image_wh, objectness=None):
        logits, deltas = self.predictor(self.feature_extractor(feats, proposals))
        return self.postprocess(logits, deltas, proposals, image_wh, objectness)

    def postprocess(self, logits, deltas, proposals, image_wh, objectness=None):      # box_head/inference.py:47-137
        c = self.c; W, H = image_wh; nc = logits.shape[1]
        prob = F.softmax(logits, -1)
        if objectness is not None and objectness.is_cuda:            # padding rows of the device-side RPN path (objectness -1): no detections from them
            prob = prob * (objectness >= 0).unsqueeze(1)
        boxes = clip_boxes(self.ops.box_decode(deltas, proposals, c.bbox_reg_weights).reshape(-1, 4), W, H).reshape(-1, nc * 4)
        # the reference loops over the classes (one nonzero + one NMS each); here: one nonzero over (proposal, class), one grouped NMS, and the
        # result put back into the reference's order (class ascending, then proposal index)
        ij = (prob[:, 1:] > c.score_thresh).nonzero()
        if not ij.shape[0]:
            return proposals.new_zeros((0, 4)), proposals.new_zeros((0,)), torch.zeros((0,), dtype=torch.int64, device=logits.device)
        i, j = ij[:, 0], ij[:, 1] + 1
        sc = prob[i, j]; bx = boxes.view(-1, nc, 4)[i, j]
        keep = self.ops.nms_grouped(bx, sc, j, c.nms)
        keep = keep[torch.sort(j[keep] * prob.shape[0] + i[keep], stable=True)[1]]
        rb, rs, rl = bx[keep], sc[keep], j[keep]
        if len(rs) > c.detections_per_img > 0:
            thresh, _ = torch.kthvalue(rs if rs.is_cuda else rs.cpu(), len(rs) - c.detections_per_img + 1)      # on the device: no D2H copy + .item() round trips before the nonzero
            keep = torch.nonzero(rs >= thresh.to(rs.device)).squeeze(1)
            rb, rs, rl = rb[keep], rs[keep], rl[keep]
        return rb, rs, rl


def _postprocess_static(self, logits, deltas, proposals, image_wh, objectness, cap):
    """postprocess() with STATIC shapes (no nonzero / .item(): capturable in a hipGraph, no host round trip): the same detections in the same order
    (class ascending, then proposal index — box_head/inference.py:96-137), written into `cap` fixed slots.  Returns (boxes [cap,4], scores [cap], labels [cap] i64,
    n_det) where n_det (device scalar) counts the detections the reference would return; when score ties at the kthvalue cut push it past `cap`, only the first
    `cap` are stored and the caller falls back to postprocess() for that image (n_det > cap).
      * candidates: class-major score matrix S[j, i] (80 x 1000), one stable descending sort per class row, below-threshold entries pushed behind (score -1);
      * NMS: one segment per class through the segmented HIP kernels (fixed 1000-slot segments, the live length per class is a device array);
      * the detections_per_img rule (kthvalue on the CPU in the reference): threshold = the cap-th largest kept score, keep >= threshold;
      * compaction: a cumulative count over the (class, proposal) grid scatters the survivors into their slots."""
    c = self.c; W, H = image_wh; nc = logits.shape[1]; N = logits.shape[0]; dev = logits.device
    prob = F.softmax(logits, -1)
    if objectness is not None:
        prob = prob * (objectness >= 0).unsqueeze(1)
    boxes = clip_boxes(self.ops.box_decode(deltas, proposals, c.bbox_reg_weights).reshape(-1, 4), W, H).reshape(N, nc, 4)
    S = prob[:, 1:].t().contiguous()                                       # [nc-1, N]
    valid = S > c.score_thresh
    key = torch.where(valid, S, S.new_full((), -1.0))
    _, order = torch.sort(key, dim=1, descending=True, stable=True)        # per class: score descending, ties by proposal index
    bx_cls = boxes[:, 1:].permute(1, 0, 2).contiguous()                    # [nc-1, N, 4]
    bx_sorted = torch.gather(bx_cls, 1, order.unsqueeze(-1).expand(-1, -1, 4)).reshape(-1, 4)
    st = getattr(self, "_static_tabs", None)
    if st is None or st[0] != (nc, N, str(dev)):
        st = ((nc, N, str(dev)), torch.arange(nc - 1, device=dev, dtype=torch.int32) * N,
              (torch.arange(1, nc, device=dev, dtype=torch.int64).unsqueeze(1).expand(nc - 1, N)).reshape(-1).contiguous())
        self._static_tabs = st
    seg_off, lab_grid = st[1], st[2]
    seg_n = valid.sum(1).to(torch.int32)
    keep, _ = self.ops.nms_segments(bx_sorted, seg_off, seg_n, N, c.nms)   # [nc-1, N] kept sorted positions, -1 padded
    kept_sorted = torch.zeros((nc - 1, N + 1), dtype=torch.bool, device=dev)
    kept_sorted.scatter_(1, torch.where(keep >= 0, keep, torch.full_like(keep, N)).long(), True)
    Fk = torch.zeros((nc - 1, N), dtype=torch.bool, device=dev).scatter_(1, order, kept_sorted[:, :N])      # back to (class, proposal)
    Sc = torch.where(Fk, S, S.new_full((), -1.0)).reshape(-1)
    n_keep = Fk.sum()
    if c.detections_per_img > 0:
        top = torch.sort(Sc, descending=True)[0]
        th = torch.where(n_keep > c.detections_per_img, top[c.detections_per_img - 1], top.new_full((), -0.5))
        Fk = Fk & (Sc.reshape(nc - 1, N) >= th)
    flat = Fk.reshape(-1)
    pos = torch.cumsum(flat, 0) - 1
    n_det = flat.sum()
    dst = torch.where(flat & (pos < cap), pos, torch.full_like(pos, cap))
    ob = boxes.new_zeros((cap + 1, 4)).index_copy_(0, dst, bx_cls.reshape(-1, 4))
    osc = boxes.new_zeros((cap + 1,)).index_copy_(0, dst, S.reshape(-1))
    olb = torch.zeros((cap + 1,), dtype=torch.int64, device=dev).index_copy_(0, dst, lab_grid)
    return ob[:cap], osc[:cap], olb[:cap], n_det


_BoxHead.postprocess_static = _postprocess_static


def _postprocess_fused(self, logits, deltas, proposals, image_wh, objectness, cap):
    """postprocess_static() with the selection logic in ordered device kernels (csrc/detpost.hip): softmax (torch) -> vido_det_class_sort (threshold + per-class descending
    order + decode + clip) -> segmented NMS -> vido_det_select (detections_per_img rule + the result lists in (class, proposal) order).  ~8 launches instead of ~45."""
    c = self.c; W, H = image_wh; nc = logits.shape[1]; N = logits.shape[0]; dev = logits.device
    prob = F.softmax(logits, -1)
    seg, order, seg_n = self.ops.det_class_sort(prob, deltas, proposals, objectness, c.score_thresh, c.bbox_reg_weights, (W, H))
    if getattr(self, "_fseg_key", None) != (nc, N, str(dev)):
        self._fseg_off = torch.arange(nc - 1, device=dev, dtype=torch.int32) * N; self._fseg_key = (nc, N, str(dev))
    keep, cnt = self.ops.nms_segments(seg, self._fseg_off, seg_n, N, c.nms)
    ob, osc, ol, nd = self.ops.det_select(prob, seg, order, keep, cnt, c.detections_per_img, cap)
    return ob, osc, ol, nd[0]


_BoxHead.postprocess_fused = _postprocess_fused


class _MaskFeatures(nn.Module):            # roi_mask_feature_extractors.py:17-65
    def __init__(self, c, ops):
        super().__init__()
        self.pooler = _Pooler(c.mask_resolution, c.pool_scales, c.sampling_ratio, ops)
        cin = c.fpn_out; self.names = []
        for i, ch in enumerate(c.mask_layers, 1):
            setattr(self, "mask_fcn%d" % i, nn.Conv2d(cin, ch, 3, 1, 1)); self.names.append("mask_fcn%d" % i); cin = ch

    def forward(self, feats, boxes):
        x = self.pooler(feats, boxes)
        for n in self.names:
            x = _conv_bias_relu(getattr(self, n), x, self.pooler.ops if hasattr(self.pooler.ops, "bias_act_") else None)
        return x


class _MaskPredictor(nn.Module):           # roi_mask_predictors.py:11-31
    def __init__(self, c):
        super().__init__()
        self.conv5_mask = nn.ConvTranspose2d(c.mask_layers[-1], c.mask_layers[-1], 2, 2, 0)
        self.mask_fcn_logits = nn.Conv2d(c.mask_layers[-1], c.num_classes, 1)

    _ops = None

    def forward(self, x):
        return self.mask_fcn_logits(_conv_bias_relu(self.conv5_mask, x, self._ops))

    def selected(self, x, labels):
        """sigmoid(forward(x))[arange(n), labels][:, None] with only each detection's own class channel computed (vido_mask_logit_select): 1 launch instead of the 81-channel
        convolution, its bias pass, the sigmoid and the gather."""
        y = _conv_bias_relu(self.conv5_mask, x, self._ops)
        if self._ops is not None and y.is_cuda and hasattr(self._ops, "mask_logit_select") and not os.environ.get("VIDO_NO_MASK_SELECT") and labels.dtype == torch.int64:
            return self._ops.mask_logit_select(y.contiguous(), self.mask_fcn_logits, labels.contiguous())
        return self.mask_fcn_logits(y).sigmoid()[torch.arange(x.shape[0], device=labels.device), labels][:, None]


class _MaskHead(nn.Module):
    buckets = (4, 8, 16, 32, 64, 100)

    def __init__(self, c, ops):
        super().__init__()
        self.feature_extractor = _MaskFeatures(c, ops); self.predictor = _MaskPredictor(c)

    def forward(self, feats, boxes, labels):                         # mask_head/inference.py:29-47: per-detection class channel
        if not len(boxes):
            return feats[0].new_zeros((0, 1, 2 * self.feature_extractor.pooler.res, 2 * self.feature_extractor.pooler.res))
        n = len(boxes); self.last_n = n                              # detections the mask head ran on (bench: per_frame_counts)
        idx = torch.arange(n, device=labels.device)
        if not (boxes.is_cuda and self.buckets):
            return self.predictor(self.feature_extractor(feats, boxes)).sigmoid()[idx, labels][:, None]
        # MIOpen compiles / selects its kernels per problem shape, and the detection count is the batch dimension of the mask head's six convolutions: run it in
        # chunks of at most buckets[-1] detections, each rounded up to one of a few fixed sizes (zero-area padding boxes, their rows dropped again), so that no
        # detection count ever means new kernels (score ties at the kthvalue cut can push the count past detections_per_img, box_head/inference.py:131-137)
        out = []; big = self.buckets[-1]
        for a in range(0, n, big):
            bx, lb = boxes[a:a + big], labels[a:a + big]; k = len(bx)
            m = next(b for b in self.buckets if b >= k)
            if m > k:
                bx = torch.cat([bx, bx.new_zeros((m - k, 4))]); lb = torch.cat([lb, lb.new_zeros((m - k,))])
            g = self.graphed.get(m) if self.graphed and feats[0].data_ptr() == self._graph_feat_ptr else None
            logits = g(bx) if g is not None else self.chunk_logits(feats, bx)          # hipGraph per bucket (pipeline.NetNodes): ~40 launches -> one replay
            out.append(logits.sigmoid()[torch.arange(m, device=lb.device), lb][:k, None])
        return torch.cat(out)

    graphed = None; _graph_feat_ptr = 0; last_n = 0

    def chunk_logits(self, feats, bx):
        return self.predictor(self.feature_extractor(feats, bx))

    def capture_buckets(self, feats, graphed_cls):
        """One captured graph per bucket size over the STATIC feature maps `feats` (the outputs of the captured trunk: same addresses every frame)."""
        feats = list(feats)
        self.graphed = {}
        for b in self.buckets:
            ex = torch.tensor([[10.0, 10.0, 200.0, 300.0]], device=feats[0].device).repeat(b, 1)
            self.graphed[b] = graphed_cls(lambda bx, _f=feats: self.chunk_logits(_f, bx), [ex])
        self._graph_feat_ptr = feats[0].data_ptr()


class _RoiHeads(nn.Module):
    def __init__(self, c, ops):
        super().__init__()
        self.box = _BoxHead(c, ops); self.mask = _MaskHead(c, ops)


def paste_masks(masks, boxes, im_h, im_w, thresh=0.5, padding=1):
    """Masker / paste_mask_in_image (mask_head/inference.py:87-160): masks [n,1,M,M] probabilities, boxes [n,4] in the target
    image -> bool [n, im_h, im_w].  Per detection: pad the mask by 1 px, scale the box by (M+2)/M, truncate to int,
    bilinear-resize to the box size, threshold, crop to the image."""
    n, M = masks.shape[0], masks.shape[-1]
    out = torch.zeros((n, im_h, im_w), dtype=torch.bool, device=masks.device)
    if n == 0:
        return out
    scale = float(M + 2 * padding) / M
    padded = F.pad(masks.float(), (padding,) * 4)
    b = boxes.float()
    wh, hh = (b[:, 2] - b[:, 0]) * 0.5 * scale, (b[:, 3] - b[:, 1]) * 0.5 * scale
    xc, yc = (b[:, 2] + b[:, 0]) * 0.5, (b[:, 3] + b[:, 1]) * 0.5
    ib = torch.stack([xc - wh, yc - hh, xc + wh, yc + hh], 1).to(torch.int32).cpu().tolist()
    for i, (x0, y0, x1, y1) in enumerate(ib):
        w, h = max(x1 - x0 + 1, 1), max(y1 - y0 + 1, 1)
        m = F.interpolate(padded[i:i + 1], size=(h, w), mode="bilinear", align_corners=False)[0, 0] > thresh
        xa, xb, ya, yb = max(x0, 0), min(x1 + 1, im_w), max(y0, 0), min(y1 + 1, im_h)
        if xb > xa and yb > ya:
            out[i, ya:yb, xa:xb] = m[ya - y0:yb - y0, xa - x0:xb - x0]
    return out


class MaskRCNN(nn.Module):
    """GeneralizedRCNN (modeling/detector/generalized_rcnn.py:33-65), eval mode, MASK_ON."""

    def __init__(self, ops, config=None):
        super().__init__()
        self.config = c = config or MaskRCNNConfig()
        self.backbone = nn.Sequential(OrderedDict([("body", _Body(c)), ("fpn", _FPN(c))]))
        self.rpn = _RPN(c, ops)
        self.roi_heads = _RoiHeads(c, ops)
        for m in self.modules():
            if isinstance(m, (_RPNHead, _MaskPredictor, _FPN)):
                m._ops = ops if hasattr(ops, "bias_act_") else None

    @torch.no_grad()
    def forward(self, image):
        """image: [1,3,H,W] float (0..255, the reference feeds unnormalised RGB).  Returns dict(boxes, scores, labels, masks[n,1,28,28])."""
        feats, logits, deltas = self.trunk(image)
        return self.heads(feats, logits, deltas, image.shape[-2:])

    @torch.no_grad()
    def trunk(self, image):
        """The static-shape part (backbone + FPN + RPN head convolutions): everything before the first data-dependent shape.  nets/fuse.py::Graphed
        captures it into one hipGraph."""
        feats = self.backbone(image)
        logits, deltas = self.rpn.head(feats)
        return feats, logits, deltas

    @torch.no_grad()
    def heads(self, feats, logits, deltas, image_hw):
        H, W = image_hw
        proposals, objectness = self.rpn.proposals(feats, logits, deltas, (W, H))
        maps = self.fpn_maps(feats)
        boxes, scores, labels = self.roi_heads.box(maps, proposals, (W, H), objectness)
        masks = self.roi_heads.mask(maps, boxes, labels)
        # proposals / objectness: the device-side RPN path returns fixed-size lists whose trailing rows (objectness -1) are padding; n_proposals counts the real ones
        return dict(boxes=boxes, scores=scores, labels=labels, masks=masks, proposals=proposals, objectness=objectness, n_proposals=(objectness >= 0).sum())


def _fpn_maps(self, feats):
    """The pooled levels + their channels-last copies (one HIP transpose per level and frame, shared by the box and the mask pooler)."""
    ops = self.rpn.ops; lv = tuple(feats[:len(self.config.pool_scales)])
    if not (hasattr(ops, "to_nhwc") and lv[0].is_cuda and len(lv) == 4):
        return lv
    m = FpnMaps(lv); m.nhwc = tuple(ops.to_nhwc(f) for f in lv)
    return m


@torch.no_grad()
def _heads_static(self, feats, logits, deltas, image_hw, cap=None):
    """heads() with static shapes end to end (device-side RPN selection, fixed 1000 proposals, postprocess_static, the mask head on `cap` padded slots): no host
    synchronisation anywhere, so trunk + heads + label image replay as ONE hipGraph.  Slots >= n_det hold zero boxes / label 0."""
    H, W = image_hw; c = self.config; cap = cap or c.detections_per_img
    fused = self.fused_post and hasattr(self.rpn.ops, "det_select") and c.pre_nms_top_n <= 1024 and c.rpn_min_size <= 0       # csrc/detpost.hip (False: the torch-op form of round 3's first static head)
    proposals, objectness = (self.rpn.proposals_fused if fused else self.rpn.proposals_device)(feats, logits, deltas, (W, H))
    maps = self.fpn_maps(feats)
    bh = self.roi_heads.box
    lg, dl = bh.predictor(bh.feature_extractor(maps, proposals))
    boxes, scores, labels, n_det = (bh.postprocess_fused if fused else bh.postprocess_static)(lg, dl, proposals, (W, H), objectness, cap)
    mh = self.roi_heads.mask
    masks = mh.predictor.selected(mh.feature_extractor(maps, boxes), labels)      # (only each slot's own class channel: vido_mask_logit_select)
    return dict(boxes=boxes, scores=scores, labels=labels, masks=masks, n_det=n_det, proposals=proposals, objectness=objectness, n_proposals=(objectness >= 0).sum())


MaskRCNN.fused_post = True
MaskRCNN.fpn_maps = _fpn_maps
MaskRCNN.heads_static = _heads_static


@torch.no_grad()
def analyse_image_static(net, feats, logits, deltas, out_hw, feed=(1088, 800), confidence=0.8, cap=None):
    """analyse_image's tail on the static head: (label image [H,W] u8, labels [cap] i64 in descending score order with 0 for unused slots, n_labels, n_det) — all device
    tensors, nothing synchronised.  Detections that fail the confidence test (and the padding slots) keep their slot with class index 0: they add nothing to the label image
    (run_mask_rcnn.py:112-118 sums mask * class_index), which is what selecting them away does in the reference.  n_det > cap: the caller must redo the image with analyse_image()."""
    H, W = out_hw
    out = net.heads_static(feats, logits, deltas, feed, cap)
    cap = out["boxes"].shape[0]
    rw, rh = float(W) / feed[1], float(H) / feed[0]
    cache = net.__dict__.setdefault("_const_cache", {})            # constants are made once, outside any capture: tensor-from-list is a synchronous host-to-device copy, which a hipGraph capture forbids
    key = ("box_scale", rw, rh, str(out["boxes"].device))
    if key not in cache:
        cache[key] = out["boxes"].new_tensor([rw, rh, rw, rh])
    boxes = out["boxes"] * cache[key]
    ops = net.rpn.ops
    nd = out["n_det"]
    if hasattr(ops, "det_order") and cap <= 1024 and nd.dtype == torch.int32 and out["labels"].dtype == torch.int64 and not os.environ.get("VIDO_NO_DET_ORDER"):
        order, labels, n_live = ops.det_order(out["scores"].contiguous(), out["labels"].contiguous(), nd.reshape(1), confidence)      # one launch instead of eleven
    else:
        live = (out["scores"] > confidence) & (torch.arange(cap, device=boxes.device) < nd)
        order = torch.sort(torch.where(live, out["scores"], out["scores"].new_full((), -1.0)), descending=True, stable=True)[1]
        labels = torch.where(live, out["labels"], torch.zeros_like(out["labels"]))[order]
        n_live = live.sum()
    img = ops.mask_label_image(out["masks"][order], boxes[order], labels, H, W)
    return img, labels, n_live, out["n_det"]


def image_to_feed(bgr, dev, feed=(1088, 800), ops=None):
    """predictor.py:267-283: HxWx3 u8 BGR -> [1,3,feed_h,feed_w] float RGB, area-resized, not normalised."""
    if ops is not None and torch.is_tensor(bgr) and bgr.is_cuda and bgr.dtype == torch.uint8:
        return ops.area_feed(bgr.contiguous(), feed)                 # one HIP pass (vido_area_feed), identical values
    t = (bgr.to(dev).flip(-1) if torch.is_tensor(bgr) else torch.as_tensor(bgr[:, :, ::-1].copy(), device=dev)).permute(2, 0, 1).float().unsqueeze(0)
    return F.interpolate(t, size=feed, mode="area")


@torch.no_grad()
def analyse_image(net, bgr, feed=(1088, 800), confidence=0.8, trunk=None):
    """predictor.py:compute_prediction + select_top_predictions (:215-283) and run_mask_rcnn.py:create_pixel_masks (:83-123):
    HxWx3 u8 BGR -> (label image HxW u8 = sum of mask * class index, label indices).  The frame is area-resized to 800x1088
    (W x H, cv2.INTER_AREA; third-party, restated with torch's area interpolation), flipped to RGB, NOT normalised."""
    dev = next(net.parameters()).device
    H, W = bgr.shape[:2]
    if trunk is not None:                                    # hipGraph-captured static part (pipeline.NetNodes): u8 HxWx3 BGR in -> (feats, logits, deltas)
        feats, logits, deltas = trunk(bgr)
        out = net.heads(feats, logits, deltas, feed)
    else:
        out = net(image_to_feed(bgr, dev, feed))
    rw, rh = float(W) / feed[1], float(H) / feed[0]
    boxes = out["boxes"] * out["boxes"].new_tensor([rw, rh, rw, rh]) if rw != rh else out["boxes"] * rw
    keep = torch.nonzero(out["scores"] > confidence).squeeze(1)
    keep = keep[out["scores"][keep].sort(0, descending=True)[1]]
    labels = out["labels"][keep]
    ops = getattr(net.rpn, "ops", None)
    if hasattr(ops, "mask_label_image") and out["masks"].is_cuda:         # Masker + label image in one HIP pass (no per-detection host loop)
        return ops.mask_label_image(out["masks"][keep], boxes[keep], labels, H, W), labels
    pasted_kept = paste_masks(out["masks"][keep], boxes[keep], H, W)      # the reference pastes every detection and then selects; only the selected ones reach the output
    # label image = sum over detections of mask * class index, accumulated in u8 (wraps on overlap, like the reference's numpy loop);
    # one reduction on the device instead of a host-synchronising loop over the detections
    if len(keep):
        img = (pasted_kept.to(torch.int32) * labels.view(-1, 1, 1).to(torch.int32)).sum(0).remainder(256).to(torch.uint8)
    else:
        img = torch.zeros((H, W), dtype=torch.uint8, device=dev)
    return img, labels
