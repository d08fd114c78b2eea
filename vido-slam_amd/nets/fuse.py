"""Inference-time graph rewrites of the three network nodes (fp32 in, fp32 out — same arithmetic type as the reference nodes).

* fold_batchnorm(net, ops): every (convolution, frozen / eval-mode batch norm) pair becomes one convolution — the norm's per-channel
  scale goes into the convolution weights, its shift becomes a bias (maskrcnn_benchmark/layers/batch_norm.py:19-31 is an affine map with
  constants; torchvision's eval-mode BatchNorm2d in mono_depth2/src/networks/resnet_encoder.py:62-98 likewise) — and the bias add, the
  shortcut add and the ReLU that follow run as ONE in-place HIP pass (vido_bias_res_act) instead of 3-4 elementwise launches.
  The state dict is untouched (the reference's checkpoints still load); the folded tensors are derived buffers, refreshed by calling
  fold_batchnorm again after loading new weights.
* Graphed: a static-shape forward captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed per frame: one launch per network
  instead of several hundred (the nodes run at batch 1, where the host launch rate — not the GPU — paces the small layers).
"""
import os
import torch
import torch.nn as nn
import torch.nn.functional as F


def _fold(conv, bn, eps):
    """(weight', bias') of conv followed by y = (x - mean) * w * rsqrt(var + eps) + b."""
    scale = bn.weight * (bn.running_var + eps).rsqrt()
    w = conv.weight * scale.reshape(-1, 1, 1, 1)
    b = bn.bias - bn.running_mean * scale
    if conv.bias is not None:
        b = b + conv.bias * scale
    return w.detach().contiguous(), b.detach().contiguous()


def _eps(bn):
    return float(getattr(bn, "eps", 0.0))              # FrozenBatchNorm2d has no epsilon (batch_norm.py:24)


def fold_batchnorm(net, ops):
    """Installs the folded fast path on every block that has one (`fused_forward`); returns the number of folded pairs.  `ops` is a HipOps
    (bias_res_act_): there is no CPU flavour of this path."""
    from .maskrcnn import _Bottleneck, _Stem
    from .monodepth2 import _Basic, ResnetEncoder18, DepthDecoder
    n = 0
    for m in net.modules():
        if isinstance(m, _Bottleneck):
            m._w1, m._b1 = _fold(m.conv1, m.bn1, _eps(m.bn1)); m._w2, m._b2 = _fold(m.conv2, m.bn2, _eps(m.bn2)); m._w3, m._b3 = _fold(m.conv3, m.bn3, _eps(m.bn3)); n += 3
            if m.downsample is not None:
                m._wd, m._bd = _fold(m.downsample[0], m.downsample[1], _eps(m.downsample[1])); n += 1
            m._ep = ops.bias_res_act_
            # the grouped 3x3 convolution + its bias + ReLU as one matrix-core kernel (csrc/gconv.hip) where it has a form for the layer
            c2 = m.conv2; m._w2p = None; m._ops = ops
            if c2.groups > 1 and tuple(c2.stride) == (1, 1) and tuple(c2.padding) == (1, 1) and tuple(c2.dilation) == (1, 1) and not os.environ.get("VIDO_NO_GCONV"):
                from .ops import pack_gconv3x3
                m._w2p = pack_gconv3x3(m._w2, c2.groups)
            elif (c2.groups > 1 and tuple(c2.stride) == (2, 2) and tuple(c2.padding) == (1, 1) and tuple(c2.dilation) == (1, 1) and ((c2.out_channels // c2.groups) % 32 == 0 or (c2.out_channels // c2.groups) in (8, 16))
                  and not os.environ.get("VIDO_NO_GCONV") and not os.environ.get("VIDO_NO_GCONV_S2")):
                from .ops import pack_gconv3x3
                m._w2p = pack_gconv3x3(m._w2, c2.groups)      # the strided conv2 of a stage's first block: csrc/gconv.hip::k_gconv3x3_s2_m32 (same operand order)
            # the 1x1 convolutions as our own fp32 matrix-core GEMM with the bias (+ shortcut) + ReLU in its epilogue (csrc/conv1x1.hip).  Measured on the detector's shapes
            # (tools/prof_conv1x1.py, profiles/r4/conv1x1_microbench_v3.txt) the GEMM alone runs where the library's does (66 us / 107 TFLOP/s at 1024 -> 1024 on 50 x 68 against
            # 66-68); what it saves is the pass behind the library's: conv3's bias + shortcut + ReLU (8-13 us per block), the stride-1 shortcut's bias, and conv1's bias + ReLU,
            # which otherwise ride on conv2's operand reads as vector instructions inside ITS matrix loop (detector 9.43 -> 9.29 ms).  VIDO_CONV1X1=conv3 keeps conv1 / the
            # shortcut with the library, VIDO_NO_CONV1X1=1 everything.
            m._w1p = m._w3p = m._wdp = None; m._c1x1_min_tiles = 0
            mode = os.environ.get("VIDO_CONV1X1", "all")
            if not os.environ.get("VIDO_NO_CONV1X1") and hasattr(ops, "conv1x1_bias_act"):
                from .ops import PackedConv1x1
                m._w3p = PackedConv1x1.make(m._w3)
                if mode == "all":
                    if tuple(m.conv1.stride) == (1, 1): m._w1p = PackedConv1x1.make(m._w1)
                    if m.downsample is not None and tuple(m.downsample[0].stride) in ((1, 1), (2, 2)): m._wdp = PackedConv1x1.make(m._wd)      # (stride 2: on the subsampled map)
        elif isinstance(m, _Stem):
            m._w1, m._b1 = _fold(m.conv1, m.bn1, _eps(m.bn1)); m._ep = ops.bias_res_act_; n += 1
        elif isinstance(m, _Basic):
            m._w1, m._b1 = _fold(m.conv1, m.bn1, m.bn1.eps); m._w2, m._b2 = _fold(m.conv2, m.bn2, m.bn2.eps); n += 2
            if m.downsample is not None:
                m._wd, m._bd = _fold(m.downsample[0], m.downsample[1], m.downsample[1].eps); n += 1
            m._ep = ops.bias_res_act_
        elif isinstance(m, DepthDecoder):
            m._ops = ops if hasattr(ops, "upcat_reflect") and not os.environ.get("VIDO_NO_DEPTH_FUSED") else None
        elif isinstance(m, ResnetEncoder18):
            e = m.encoder
            w, b = _fold(e.conv1, e.bn1, e.bn1.eps)
            # (image - 0.45) / 0.225 folded in as well: conv(w, (x - m) / s) = conv(w / s, x) - sum(w) * m / s   (resnet_encoder.py:89)
            # zero padding of the ORIGINAL graph pads the normalised image with 0, i.e. the raw image with 0.45: keep the normalisation outside
            m._w1, m._b1 = w, b; m._ep = ops.bias_res_act_; n += 1
    return n


class Graphed:
    """fn(*tensors) -> tensor | tuple of tensors, all shapes static: warmed up, captured once, replayed per call on the CURRENT stream.
    Inputs are copied into the capture's static buffers; the returned tensors are the capture's static outputs (valid until the next call)."""

    def __init__(self, fn, example_inputs, warmup=2):
        self.fn = fn
        self.static_in = [x.clone() for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)
        torch.cuda.synchronize()

    def __call__(self, *inputs):
        for d, s in zip(self.static_in, inputs):
            if d.data_ptr() != s.data_ptr():
                d.copy_(s, non_blocking=True)
        self.graph.replay()
        return self.static_out
