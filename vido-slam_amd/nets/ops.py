"""torch <-> C-ABI glue for the network ops.  Device tensors are handed to the library as raw pointers
(on_device=1); the ctx adopts torch's current stream (vido_set_stream) so launches are ordered with the
surrounding torch kernels without any synchronisation."""
import ctypes as C
import os
import torch

_WINO_MIN_WGS = int(os.environ.get("VIDO_WINO_MIN_WGS", "0"))
_CONV3X3_H_MIN_WGS = 0 if os.environ.get("VIDO_CONV3X3_H") == "0" else int(os.environ.get("VIDO_CONV3X3_H_MIN_WGS", "128"))      # direct split-fp16 3x3 (csrc/conv3x3h.hip) from this many workgroups on
# which layers conv_direct_conv takes: "all", or "novalu" (default) = not the 7x7 stem / stride-2 3x3 layers, which the library runs as Winograd on the VECTOR ALUs — beside
# the detector (whose convolutions saturate the MATRIX pipe) those run in its shadow, while the direct kernel competes for the matrix pipe.  Measured (two pairs of 100
# steps, profiles/r5/convdirect_ab.txt): headline 90.3 frames/s without the direct kernel, 89.4 with it on every layer (LiteFlowNet alone 3.85 -> 3.65 ms), 90.6 with
# "novalu" (3.76 ms alone).  Leaving only the two miopenSp3AsmConv layers (stem, 32 -> 32 stride 2) to the library and taking the other stride-2 layers (implicit GEMM
# in the library) gives 90.0 against 90.4: the direct kernel's matrix-pipe time costs the frame more than the library's kernels there too.  The trade does NOT extend to
# the dense 3x3 layers: LiteFlowNet on the library's vector-ALU Winograd throughout gives 82 frames/s.
# Round 6 (detector in split fp16, profiles/r6/convdirect_set_ab.txt): 125.0 frames/s with "novalu" against 124.3 / 124.5 with "all" (the chain without the detector: 191.4 against 193.8).
_CONVDIRECT_SET = os.environ.get("VIDO_CONVDIRECT_SET", "novalu")


def correlation_torch_reference(first, second, stride):
    """Plain-PyTorch fp32 reference of the 7x7 cost volume (used by the CPU tests only; the product path is the HIP
    kernel): out[b,(p+3)*7+(o+3),y,x] = mean_c f1[b,c,y*s,x*s] * f2[b,c,(y+p)*s,(x+o)*s]."""
    a, b = first[:, :, ::stride, ::stride], second[:, :, ::stride, ::stride]
    pad = torch.nn.functional.pad(b, (3, 3, 3, 3))
    H, W = a.shape[2], a.shape[3]
    return torch.stack([(a * pad[:, :, p:p + H, o:o + W]).mean(1) for p in range(7) for o in range(7)], 1)


class HipOps:
    """FunctionCorrelation / ROIAlign / nms / box decode on device tensors through libvido_slam_hip.so."""

    def __init__(self, ctx):
        self.ctx = ctx

    def _adopt_stream(self):
        st = torch.cuda.current_stream().cuda_stream
        self.ctx._check(self.ctx.lib.vido_set_stream(self.ctx.h, C.c_void_p(st), 1))

    def correlation(self, first, second, stride):
        if not first.is_cuda:
            raise RuntimeError("HipOps.correlation needs CUDA(HIP) tensors; there is no CPU fallback")
        first = first.contiguous().float(); second = second.contiguous().float()
        B, Cc, H, W = first.shape
        out = torch.empty((B, 49, (H + stride - 1) // stride, (W + stride - 1) // stride), device=first.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_correlation(self.ctx.h, C.c_void_p(first.data_ptr()), C.c_void_p(second.data_ptr()), B, Cc, H, W, stride,
                                                      C.c_void_p(out.data_ptr()), 1))
        return out

    def bias_act_(self, x, bias, slope):
        """in place: x = leaky_relu(x + bias[None, :, None, None], slope)"""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        N, Cc, H, W = x.shape
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()), N, Cc, H, W, C.c_float(slope)))
        return x

    def deconv2x2_conv(self, conv, x, slope):
        """nn.ConvTranspose2d `conv` (2 x 2, stride 2, no padding) + bias + activation of a batch as one split-fp16 GEMM with a scatter epilogue (csrc/conv1x1.hip, RES 3), else
        None; the packed weight is cached on the module."""
        w = conv.weight
        if (tuple(w.shape[2:]) != (2, 2) or tuple(conv.stride) != (2, 2) or tuple(conv.padding) != (0, 0) or tuple(conv.output_padding) != (0, 0) or tuple(conv.dilation) != (1, 1)
                or conv.groups != 1 or not x.is_cuda or x.dtype != torch.float32 or os.environ.get("VIDO_NO_DECONV_H")):
            return None
        n, cin, H, W = (int(v) for v in x.shape); cout = int(w.shape[1])
        if not self.ctx.lib.vido_deconv2x2_supported(n, cin, cout, H, W):
            return None
        key = (w.data_ptr(), w._version, str(x.device))
        if getattr(conv, "_dc_key", None) != key:
            conv._dc_w = pack_deconv2x2(w).to(x.device); conv._dc_key = key
        out = torch.empty((n, cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * n * cin * cout * 4 * H * W
        self._adopt_stream()
        b = conv.bias
        self.ctx._check(self.ctx.lib.vido_deconv2x2_bias_act(self.ctx.h, C.c_void_p(x.contiguous().data_ptr()), C.c_void_p(conv._dc_w.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                                             C.c_void_p(out.data_ptr()), n, cin, cout, H, W, C.c_float(slope)))
        return out

    def det_order(self, scores, labels, n_det, confidence):
        """(order int64 [cap], labels in that order with 0 for slots that fail `scores > confidence and slot < n_det`, number of live slots) — analyse_image_static's tail
        in one launch (csrc/nets.hip::k_det_order); the order is torch.sort(where(live, scores, -1), descending=True, stable=True)."""
        assert scores.is_cuda and scores.dtype == torch.float32 and labels.dtype == torch.int64 and n_det.dtype == torch.int32 and scores.is_contiguous() and labels.is_contiguous()
        cap = int(scores.shape[0]); dev = scores.device
        order = torch.empty((cap,), device=dev, dtype=torch.int64); lab = torch.empty((cap,), device=dev, dtype=torch.int64); n_live = torch.empty((), device=dev, dtype=torch.int64)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_det_order(self.ctx.h, C.c_void_p(scores.data_ptr()), C.c_void_p(labels.data_ptr()), C.c_void_p(n_det.data_ptr()), C.c_float(confidence), cap,
                                                    C.c_void_p(order.data_ptr()), C.c_void_p(lab.data_ptr()), C.c_void_p(n_live.data_ptr())))
        return order, lab, n_live

    def roi_levels(self, boxes, k_min, k_max):
        """LevelMapper of the FPN pooler for boxes [n, 4] f32 -> int32 [n] in 0 .. k_max - k_min, one launch (csrc/nets.hip::k_roi_levels: the torch expression's fp32 operations
        in the same order)."""
        assert boxes.is_cuda and boxes.dtype == torch.float32
        boxes = boxes.contiguous(); n = int(boxes.shape[0])
        out = torch.empty((n,), device=boxes.device, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_roi_levels(self.ctx.h, C.c_void_p(boxes.data_ptr()), n, C.c_float(k_min), C.c_float(k_max), C.c_void_p(out.data_ptr())))
        return out

    def mask_logit_select(self, feat, conv, labels):
        """sigmoid(conv(feat))[arange(n), labels][:, None] for a 1x1 `conv` (the mask head's logits layer) computing only each detection's own class channel (csrc/nets.hip)."""
        assert feat.is_cuda and feat.is_contiguous() and feat.dtype == torch.float32 and labels.dtype == torch.int64 and labels.is_contiguous()
        n, c, H, W = feat.shape; classes = int(conv.weight.shape[0])
        w = conv.weight.reshape(classes, c)
        if not w.is_contiguous():
            w = w.contiguous()
        out = torch.empty((n, 1, H, W), device=feat.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_mask_logit_select(self.ctx.h, C.c_void_p(feat.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(conv.bias.data_ptr()) if conv.bias is not None else None,
                                                            C.c_void_p(labels.data_ptr()), C.c_void_p(out.data_ptr()), int(n), int(c), int(H * W), classes))
        return out

    def bias_res_act_(self, x, bias, res, slope):
        """in place: x = leaky_relu(x + bias[None, :, None, None] + res, slope); res None -> bias_act_ (slope 0 = ReLU, 1 = none)"""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        N, Cc, H, W = x.shape
        if res is not None:
            assert res.shape == x.shape and res.is_contiguous() and res.dtype == torch.float32
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_bias_res_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()),
                                                       C.c_void_p(res.data_ptr()) if res is not None else None, N, Cc, H, W, C.c_float(slope)))
        return x

    def bias_unary_(self, x, bias, kind):
        """in place: x = f(x + bias[None, :, None, None]); kind "elu" | "sigmoid" """
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        N, Cc, H, W = x.shape
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_bias_unary(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()), N, Cc, H, W, 1 if kind == "elu" else 2))
        return x

    def upcat_reflect(self, x, skip=None):
        """ReflectionPad2d(1)(cat([interpolate(x, scale_factor=2, mode="nearest"), skip], 1)) for one image, one pass."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == 1
        _, C1, h, w = x.shape
        C2 = 0
        if skip is not None:
            assert skip.is_contiguous() and skip.shape[0] == 1 and tuple(skip.shape[2:]) == (2 * h, 2 * w)
            C2 = skip.shape[1]
        out = torch.empty((1, C1 + C2, 2 * h + 2, 2 * w + 2), device=x.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_upcat_reflect(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(skip.data_ptr()) if skip is not None else None, C1, C2, h, w, C.c_void_p(out.data_ptr())))
        return out

    def minmax_norm_u16(self, d):
        """((d - d.min()) / (d.max() - d.min() + 1e-12) * 65536).clamp(0, 65535).to(int32) with one reduction and one pass."""
        d = d.contiguous()
        mm = torch.stack(torch.aminmax(d))
        out = torch.empty(d.shape, device=d.device, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_minmax_norm_u16(self.ctx.h, C.c_void_p(d.data_ptr()), C.c_void_p(mm.data_ptr()), C.c_int64(d.numel()), C.c_void_p(out.data_ptr())))
        return out

    def area_feed(self, bgr, feed, div=1.0):
        """u8 HxWx3 BGR device tensor -> f32 [1,3,feed_h,feed_w] RGB, area-resized, / div (flip + permute + float + interpolate(area) + div in one pass)"""
        assert bgr.is_cuda and bgr.dtype == torch.uint8 and bgr.is_contiguous() and bgr.dim() == 3 and bgr.shape[2] == 3
        out = torch.empty((1, 3, int(feed[0]), int(feed[1])), device=bgr.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_area_feed(self.ctx.h, C.c_void_p(bgr.data_ptr()), bgr.shape[0], bgr.shape[1], C.c_void_p(out.data_ptr()), int(feed[0]), int(feed[1]), C.c_float(div)))
        return out

    def backwarp(self, x, flow):
        """layers.py Backward: bilinear warp of x by flow (pixels), zero padding"""
        assert x.is_cuda and x.dtype == torch.float32 and flow.dtype == torch.float32 and x.shape[0] == flow.shape[0] and x.shape[2:] == flow.shape[2:] and flow.shape[1] == 2
        x = x.contiguous(); flow = flow.contiguous()
        out = torch.empty_like(x)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_backwarp(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(flow.data_ptr()), x.shape[0], x.shape[1], x.shape[2], x.shape[3], C.c_void_p(out.data_ptr())))
        return out

    def deconv4s2_depthwise(self, x, weight, input_slope=1.0):
        """ConvTranspose2d(C, C, 4, 2, 1, groups=C, bias=False)(leaky_relu(x, input_slope)) in one HIP pass."""
        x = x.contiguous(); B, Cc, H, W = x.shape
        assert weight.shape == (Cc, 1, 4, 4) and weight.is_contiguous()
        out = torch.empty((B, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_deconv4s2_depthwise(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(weight.data_ptr()), B, Cc, H, W, C.c_float(input_slope), C.c_void_p(out.data_ptr())))
        return out

    def gconv3x3_supported(self, H, W, cpg_in, cpg_out):
        return bool(self.ctx.lib.vido_gconv3x3_supported(int(H), int(W), int(cpg_in), int(cpg_out)))

    def gconv3x3_bias_act(self, x, w_packed, bias, groups, slope=0.0, in_bias=None):
        """leaky_relu(conv2d(x', w, None, 1, 1, 1, groups) + bias, slope) for one image as one matrix-core launch; w_packed = pack_gconv3x3(w, groups);
        x' = x, or relu(x + in_bias[None, :, None, None]) when in_bias is given."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == 1
        _, Cin, H, W = x.shape
        cpg_in = Cin // groups; cpg_out = bias.numel() // groups
        out = torch.empty((1, bias.numel(), H, W), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * bias.numel() * cpg_in * 9 * H * W      # (torch's FlopCounterMode does not see this launch; bench.py adds it)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_gconv3x3_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(in_bias.data_ptr()) if in_bias is not None else None,
                                                            C.c_void_p(w_packed.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()),
                                                            int(groups), cpg_in, cpg_out, H, W, C.c_float(slope)))
        return out

    def gconv3x3_s2_supported(self, H, W, cpg_in, cpg_out):
        return bool(self.ctx.lib.vido_gconv3x3_s2_supported(int(H), int(W), int(cpg_in), int(cpg_out)))

    def gconv3x3_s2_bias_act(self, x, w_packed, bias, groups, slope=0.0):
        """leaky_relu(conv2d(x, w, None, 2, 1, 1, groups) + bias, slope) for one image as one matrix-core launch (csrc/gconv.hip::k_gconv3x3_s2_m32); w_packed = pack_gconv3x3(w, groups)."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == 1
        _, Cin, H, W = x.shape
        cpg_in = Cin // groups; cpg_out = bias.numel() // groups
        out = torch.empty((1, bias.numel(), (H + 1) // 2, W // 2), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * bias.numel() * cpg_in * 9 * out.shape[2] * out.shape[3]
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_gconv3x3_s2_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()),
                                                               int(groups), cpg_in, cpg_out, H, W, C.c_float(slope)))
        return out

    def conv1x1_supported(self, cin, cout, hw):
        return bool(self.ctx.lib.vido_conv1x1_supported(int(cin), int(cout), int(hw)))

    def conv1x1_set_arith(self, arith):
        """0: split-fp16 (two planes, three products; fp32-equivalent, the default), 1: the fp32 matrix instruction, 2: split-bf16 (three planes, six products; fp32's range);
        returns the previous setting (vido_conv1x1_set_arith, process-wide)."""
        return int(self.ctx.lib.vido_conv1x1_set_arith(int(arith)))

    def conv1x1_range_flag(self, reset=True):
        """non-zero when a split-fp16 launch met |x| >= 65504 since the last reset (its outputs are not valid); read after the stream has been waited for"""
        return int(self.ctx.lib.vido_conv1x1_range_flag(self.ctx.h, int(bool(reset))))

    def conv1x1_layout(self, cin, cout, hw):
        return int(self.ctx.lib.vido_conv1x1_layout(int(cin), int(cout), int(hw)))

    def conv1x1_bias_act(self, x, w_packed, bias=None, residual=None, slope=1.0, residual_up2=None):
        """leaky_relu(conv2d(x, w) + bias + residual, slope) for one image, 1x1 kernel, stride 1, as one matrix-core GEMM launch (csrc/conv1x1.hip);
        w_packed = pack_conv1x1(w, conv1x1_layout(cin, cout, H * W)).  slope 0 = ReLU, 1 = none.  residual_up2: a residual at half the resolution, added nearest-upsampled
        (the FPN's top-down sum)."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == 1
        _, cin, H, W = x.shape
        if isinstance(w_packed, PackedConv1x1):                     # packed on first use, in the tile form the library picks for this (cin, cout, H * W)
            w_packed = w_packed.get(self.conv1x1_layout(cin, w_packed.cout, H * W), x.device)
        cout = (w_packed.numel() // (2 * (cin + 1)) if w_packed.dim() == 1 else w_packed.numel() // cin // 3) if w_packed.dtype == torch.int16 else w_packed.numel() // cin
        layout = self.conv1x1_layout(cin, cout, H * W)
        assert not self.conv1x1_supported(cin, cout, H * W) or tuple(w_packed.shape) == {0: (cout // 32, cin // 8, 64, 4), 1: (cout // 16, cin // 16, 64, 4), 2: (cout // 32, cin // 16, 3, 64, 8),
                                                                                         3: (2 * cout * (cin + 1),)}[layout], \
            "conv1x1: weight packed for another form (pack_conv1x1(w, layout))"      # (an unsupported shape is refused by the library call below)
        assert 0.0 <= slope <= 1.0 and (residual is None or residual_up2 is None)
        out = torch.empty((1, cout, H, W), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * cout * cin * H * W      # (torch's FlopCounterMode does not see this launch; bench.py adds it)
        self._adopt_stream()
        pb = C.c_void_p(bias.data_ptr()) if bias is not None else None
        if residual_up2 is not None:
            residual_up2 = residual_up2.contiguous(); assert tuple(residual_up2.shape) == (1, cout, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0
            self.ctx._check(self.ctx.lib.vido_conv1x1_bias_up2_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), pb, C.c_void_p(residual_up2.data_ptr()),
                                                                   C.c_void_p(out.data_ptr()), int(cin), int(cout), int(H), int(W), C.c_float(slope)))
            return out
        if residual is not None:
            residual = residual.contiguous(); assert residual.shape == out.shape
        self.ctx._check(self.ctx.lib.vido_conv1x1_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), pb,
                                                           C.c_void_p(residual.data_ptr()) if residual is not None else None, C.c_void_p(out.data_ptr()), int(cin), int(cout), int(H * W), C.c_float(slope)))
        return out

    def conv1x1_conv(self, conv, x, slope=1.0, residual=None, residual_up2=None):
        """The 1x1 convolution `conv` (nn.Conv2d, stride 1) + bias (+ residual) + activation through conv1x1_bias_act when the layer has that form, else None; the packed weight
        is cached on the module (per tile form) and rebuilt when the weight tensor changes."""
        w = conv.weight
        if (tuple(w.shape[2:]) != (1, 1) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (0, 0) or conv.groups != 1 or not x.is_cuda or x.shape[0] != 1
                or not self.conv1x1_supported(w.shape[1], w.shape[0], x.shape[2] * x.shape[3])):
            return None
        if not conv1x1_fills_chip(w.shape[0], x.shape[2] * x.shape[3]):
            return None
        layout = self.conv1x1_layout(w.shape[1], w.shape[0], x.shape[2] * x.shape[3])
        key = (w.data_ptr(), w._version, str(x.device), layout)
        if getattr(conv, "_c1_key", None) != key:
            conv._c1_w = pack_conv1x1(w, layout).to(x.device); conv._c1_key = key
        return self.conv1x1_bias_act(x.contiguous(), conv._c1_w, conv.bias, residual, slope, residual_up2)

    def conv_kxk_c2(self, conv, x, residual=None):
        """conv(x) + residual for a k x k (3, 5, 7) stride-1 `same` convolution with TWO output channels on one image — the last layer of LiteFlowNet's flow heads — as one
        stencil launch (csrc/convsmall.hip) instead of the library's im2col + GEMM + our bias / residual pass; None when the layer is not of that form."""
        w = conv.weight
        k = int(w.shape[2])
        if (w.shape[0] != 2 or w.shape[2] != w.shape[3] or k not in (3, 5, 7) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (k // 2, k // 2) or tuple(conv.dilation) != (1, 1)
                or conv.groups != 1 or not x.is_cuda or x.shape[0] != 1 or x.dtype != torch.float32):
            return None
        x = x.contiguous(); wc = w.contiguous()
        _, cin, H, W = x.shape
        out = torch.empty((1, 2, H, W), device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = residual.contiguous(); assert residual.shape == out.shape
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * 2 * cin * k * k * H * W
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_conv_kxk_c2(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(wc.data_ptr()), C.c_void_p(conv.bias.data_ptr()) if conv.bias is not None else None,
                                                      C.c_void_p(residual.data_ptr()) if residual is not None else None, C.c_void_p(out.data_ptr()), int(cin), k, int(H), int(W)))
        return out

    def conv1x1_skinny(self, x, w_packed, bias, cout, slope=1.0, residual=None):
        """leaky_relu(conv2d(x, w) + bias + residual, slope), 1x1, one image, cin even <= 256, cout <= 256: one launch without LDS (csrc/convsmall.hip::k_conv1x1_skinny);
        w_packed = pack_conv1x1_skinny(w) on the device."""
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[0] == 1
        x = x.contiguous(); cin, H, W = int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        out = torch.empty((1, cout, H, W), device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = residual.contiguous(); assert residual.shape == out.shape
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * cout * cin * H * W
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_conv1x1_skinny(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                                         C.c_void_p(residual.data_ptr()) if residual is not None else None, C.c_void_p(out.data_ptr()), cin, int(cout), C.c_longlong(H * W), C.c_float(slope)))
        return out

    def conv1x1_skinny_conv(self, conv, x, slope=1.0):
        """The same for an nn.Conv2d (LiteFlowNet's netFeat layers); None when the layer is not of that form.  Packed weight cached on the module."""
        w = conv.weight
        cout, cin = int(w.shape[0]), int(w.shape[1])
        if (tuple(w.shape[2:]) != (1, 1) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (0, 0) or conv.groups != 1 or cin % 2 or cin > 256 or cout > 256
                or not x.is_cuda or x.shape[0] != 1 or x.dtype != torch.float32):
            return None
        key = (w.data_ptr(), w._version, str(x.device))
        if getattr(conv, "_c1s_key", None) != key:
            conv._c1s_w = pack_conv1x1_skinny(w).to(x.device); conv._c1s_key = key
        return self.conv1x1_skinny(x, conv._c1s_w, conv.bias, cout, slope)

    def wino3x3_supported(self, cin, cout, H, W):
        return bool(self.ctx.lib.vido_wino3x3_supported(int(cin), int(cout), int(H), int(W)))

    def wino3x3_form(self, N, cin, cout, H, W):
        """0: the tile form, 1: the K-split form (launches that would leave most of the chip idle) — the library's own rule, vido_wino3x3_form"""
        return int(self.ctx.lib.vido_wino3x3_form(int(N), int(cin), int(cout), int(H), int(W)))

    def wino3x3_bias_act(self, x, u_packed, bias, cout, slope=1.0, form=0):
        """leaky_relu(conv2d(x, w, None, 1, 1) + bias, slope) for a batch, 3x3 kernel, as one Winograd launch whose channel contractions run on the matrix pipe (csrc/wino.hip);
        u_packed = pack_wino3x3(w, form) on x's device.  slope 0 = ReLU, 1 = none."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and u_packed.is_cuda
        N, cin, H, W = x.shape
        out = torch.empty((N, cout, H, W), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * N * cout * cin * 9 * H * W      # direct-convolution count, as FlopCounterMode would give the library call
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_wino3x3_bias_act_form(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(u_packed.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                                                C.c_void_p(out.data_ptr()), int(N), int(cin), int(cout), int(H), int(W), C.c_float(slope), int(form)))
        return out

    def wino3x3_conv(self, conv, x, slope, weight=None, bias=None):
        """The convolution `conv` (nn.Conv2d, or the folded weight / bias given) + bias + activation through wino3x3_bias_act when the layer has that form, else None.  The
        packed weight is cached on the module and rebuilt when the weight tensor changes (a checkpoint loaded later)."""
        w = conv.weight if weight is None else weight
        b = conv.bias if bias is None and weight is None else bias
        if (tuple(w.shape[2:]) != (3, 3) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1) or tuple(conv.dilation) != (1, 1) or conv.groups != 1
                or getattr(conv, "padding_mode", "zeros") != "zeros" or not x.is_cuda or not self.wino3x3_supported(w.shape[1], w.shape[0], x.shape[2], x.shape[3])
                or 4 * x.numel() >= 1 << 30 or 4 * x.shape[0] * w.shape[0] * x.shape[2] * x.shape[3] >= 1 << 30):
            return None
        # a launch of a few dozen workgroups of the tile form (small maps) runs as long as a chip-filling one; those take the K-split form (four waves of a workgroup share
        # the input channels).  VIDO_WINO_MIN_WGS=128 restores the rule of round 4 (the library's kernels below 128 workgroups).
        if _WINO_MIN_WGS > 0 and not self.ctx.lib.vido_wino3x3_fills_chip(int(x.shape[0]), int(w.shape[0]), int(x.shape[2]), int(x.shape[3]), _WINO_MIN_WGS):
            return None
        # (round 6) chip-filling launches of >= 128-channel layers take the DIRECT kernel in split-fp16 arithmetic (csrc/conv3x3h.hip): 256 -> 256 on 200 x 272 in 216 us
        # against the Winograd kernel's 325, the mask head's 100 x 14 x 14 in 97 against 156; below ~128 workgroups one workgroup's
        # K loop is the launch's time and Winograd (its K-split form) stays faster.  VIDO_CONV3X3_H=0 keeps Winograd everywhere; VIDO_CONV3X3_H_MIN_WGS moves the threshold.
        if (_CONV3X3_H_MIN_WGS > 0 and x.dtype == torch.float32 and self.ctx.lib.vido_conv3x3_h_supported(int(x.shape[0]), int(w.shape[1]), int(w.shape[0]), int(x.shape[2]), int(x.shape[3]))
                and self.ctx.lib.vido_conv3x3_h_workgroups(int(x.shape[0]), int(w.shape[0]), int(x.shape[2]), int(x.shape[3])) >= (1 if int(w.shape[0]) == 64 else _CONV3X3_H_MIN_WGS)):
            # (64-channel layers: the direct kernel wins at every size — 128 -> 64 on 30 x 40: 18 us against the K-split Winograd form's 26; profiles/r6/conv3x3_h_direct.txt)
            key = (w.data_ptr(), w._version, str(x.device))
            if getattr(conv, "_c3h_key", None) != key:
                conv._c3h_w = pack_conv3x3_h(w).to(x.device); conv._c3h_key = key
            return self.conv3x3_h_bias_act(x.contiguous(), conv._c3h_w, b, int(w.shape[0]), slope)
        form = self.wino3x3_form(x.shape[0], w.shape[1], w.shape[0], x.shape[2], x.shape[3])
        key = (w.data_ptr(), w._version, str(x.device))
        if getattr(conv, "_wino_key", None) != key:
            conv._wino_u = {}; conv._wino_key = key                          # per form: a layer shared by several map sizes (the RPN head over P2-P6) is launched in both
        if form not in conv._wino_u:
            conv._wino_u[form] = pack_wino3x3(w, form).to(x.device)
        return self.wino3x3_bias_act(x.contiguous(), conv._wino_u[form], b, int(w.shape[0]), slope, form)

    def conv3x3_h_bias_act(self, x, w_packed, bias, cout, slope):
        """leaky_relu(conv2d(x, w, stride 1, padding 1) + bias, slope) as one direct split-fp16 launch (csrc/conv3x3h.hip); w_packed = pack_conv3x3_h(w)."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
        N, cin, H, W = x.shape
        assert w_packed.dtype == torch.int16 and w_packed.numel() == 2 * cout * ((cin + 15) // 16 * 16 * 9 + 1), "conv3x3_h: weight not packed by pack_conv3x3_h for this layer"
        out = torch.empty((N, cout, H, W), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * N * cout * cin * 9 * H * W
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_conv3x3_h_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                                             C.c_void_p(out.data_ptr()), int(N), int(cin), int(cout), int(H), int(W), C.c_float(slope)))
        return out

    def fc_h(self, x, w_packed, bias, outs, slope=1.0):
        """leaky_relu(x @ w.T + bias, slope) for x [rows, k] f32 as a split-fp16 matrix-pipe GEMM split over k (csrc/fch.hip); w_packed = pack_conv1x1(w[:, :, None, None], 3).
        None when the library does not take the shape."""
        assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 2
        rows, k = int(x.shape[0]), int(x.shape[1])
        S = int(self.ctx.lib.vido_fc_h_splitk(rows, k, int(outs)))
        if S <= 0:
            return None
        assert w_packed.dtype == torch.int16 and w_packed.numel() == 2 * outs * (k + 1), "fc_h: weight not packed (pack_conv1x1(w[:, :, None, None], 3))"
        part = torch.empty((S, rows, outs), device=x.device, dtype=torch.float32); y = torch.empty((rows, outs), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * rows * k * outs
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_fc_h(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(w_packed.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                               C.c_void_p(part.data_ptr()), C.c_void_p(y.data_ptr()), rows, k, int(outs), C.c_float(slope)))
        return y

    def fc_h_linear(self, lin, x, slope=1.0):
        """nn.Linear `lin` + activation through fc_h when the layer has that form (outputs a multiple of 128, inputs a multiple of 32 S), else None; the packed weight is cached
        on the module and rebuilt when the weight tensor changes."""
        w = lin.weight
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or os.environ.get("VIDO_NO_FC_H") or int(self.ctx.lib.vido_fc_h_splitk(int(x.shape[0]), int(w.shape[1]), int(w.shape[0]))) <= 0:
            return None
        key = (w.data_ptr(), w._version, str(x.device))
        if getattr(lin, "_fch_key", None) != key:
            lin._fch_w = pack_conv1x1(w.detach().reshape(int(w.shape[0]), int(w.shape[1]), 1, 1), 3).to(x.device); lin._fch_key = key
        return self.fc_h(x.contiguous(), lin._fch_w, lin.bias, int(w.shape[0]), slope)

    def conv_direct_conv(self, conv, x, slope):
        """The convolution `conv` (nn.Conv2d: groups 1, dilation 1, zero padding, a k x k of csrc/convdirect.hip) + bias + activation as ONE direct implicit-GEMM launch on
        the matrix pipe, else None.  The packed weight is cached on the module and rebuilt when the weight tensor changes."""
        w = conv.weight
        kh, kw = int(w.shape[2]), int(w.shape[3]); sh, sw = (int(v) for v in conv.stride); ph, pw = (int(v) for v in conv.padding)
        if _CONVDIRECT_SET == "novalu" and (kh, kw) in ((3, 3), (7, 7)) and int(w.shape[0]) >= 32:
            return None                                                       # layers the library runs as Winograd on the VECTOR ALUs (see _CONVDIRECT_SET)
        if (tuple(conv.dilation) != (1, 1) or conv.groups != 1 or getattr(conv, "padding_mode", "zeros") != "zeros" or not x.is_cuda or x.dtype != torch.float32
                or not self.ctx.lib.vido_conv_direct_supported(int(w.shape[1]), int(w.shape[0]), int(x.shape[2]), int(x.shape[3]), kh, kw, sh, sw, ph, pw)
                or 4 * x.numel() >= 1 << 30):
            return None
        N, cin, H, W = x.shape; cout = int(w.shape[0])
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        if 4 * N * cout * Ho * Wo >= 1 << 30:
            return None
        key = (w.data_ptr(), w._version, str(x.device))
        if getattr(conv, "_cd_key", None) != key:
            conv._cd_w = pack_conv_direct(w).to(x.device); conv._cd_key = key
        x = x.contiguous()
        out = torch.empty((N, cout, Ho, Wo), device=x.device, dtype=torch.float32)
        self.gconv_flops = getattr(self, "gconv_flops", 0.0) + 2.0 * N * cout * cin * kh * kw * Ho * Wo
        self._adopt_stream()
        b = conv.bias
        self.ctx._check(self.ctx.lib.vido_conv_direct_bias_act(self.ctx.h, C.c_void_p(x.data_ptr()), C.c_void_p(conv._cd_w.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                                               C.c_void_p(out.data_ptr()), int(N), int(cin), cout, int(H), int(W), kh, kw, sh, sw, ph, pw, C.c_float(slope)))
        return out

    def lfn_reg_front(self, im1, im2, flow, scale, feat):
        """Regularization.forward up to netMain's input (layers.py:236-243): torch.cat([sqrt(sum((im1 - Backward(im2, flow * scale))^2)), flow - mean(flow), feat], 1) with the
        first three channels from one HIP pass."""
        im1 = im1.contiguous(); im2 = im2.contiguous(); flow = flow.contiguous()
        B, Cc, H, W = im1.shape
        out = torch.empty((B, 3 + feat.shape[1], H, W), device=im1.device, dtype=torch.float32)
        mean = flow.flatten(2).mean(2).contiguous()
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_lfn_reg_front(self.ctx.h, C.c_void_p(im1.data_ptr()), C.c_void_p(im2.data_ptr()), C.c_void_p(flow.data_ptr()), C.c_void_p(mean.data_ptr()),
                                                        C.c_float(scale), B, Cc, H, W, C.c_void_p(out.data_ptr()), out.shape[1]))
        out[:, 3:] = feat
        return out

    def lfn_reg_tail(self, dist, flow, scale_x, scale_y, k):
        """Regularization.forward after netDist (layers.py:245-262); scale_x / scale_y: the netScaleX / netScaleY modules (Conv2d(k*k, 1, 1))."""
        dist = dist.contiguous(); flow = flow.contiguous()
        B, nd, H, W = dist.shape
        assert nd == k * k
        out = torch.empty((B, 2, H, W), device=dist.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_lfn_reg_tail(self.ctx.h, C.c_void_p(dist.data_ptr()), C.c_void_p(flow.data_ptr()), C.c_void_p(scale_x.weight.data_ptr()), C.c_void_p(scale_x.bias.data_ptr()),
                                                       C.c_void_p(scale_y.weight.data_ptr()), C.c_void_p(scale_y.bias.data_ptr()), B, int(k), H, W, C.c_void_p(out.data_ptr())))
        return out

    def roi_align(self, feat, rois, output_size, spatial_scale, sampling_ratio):
        if not feat.is_cuda:
            raise RuntimeError("HipOps.roi_align needs CUDA(HIP) tensors; there is no CPU fallback")
        feat = feat.contiguous().float(); rois = rois.contiguous().float()
        B, Cc, H, W = feat.shape; ph, pw = output_size
        out = torch.empty((rois.shape[0], Cc, ph, pw), device=feat.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_roi_align(self.ctx.h, C.c_void_p(feat.data_ptr()), B, Cc, H, W, C.c_void_p(rois.data_ptr()), rois.shape[0],
                                                    C.c_float(spatial_scale), ph, pw, sampling_ratio, C.c_void_p(out.data_ptr()), 1))
        return out

    def box_decode(self, deltas, boxes, weights):
        """BoxCoder(weights).decode(deltas [n,4k], boxes [n,4]) -> [n,4k]."""
        deltas = deltas.contiguous().float(); boxes = boxes.contiguous().float(); n = deltas.shape[0]
        out = torch.empty_like(deltas)
        if n:
            w = (C.c_float * 4)(*[float(x) for x in weights])
            self._adopt_stream()
            self.ctx._check(self.ctx.lib.vido_box_decode(self.ctx.h, C.c_void_p(deltas.data_ptr()), C.c_void_p(boxes.data_ptr()), n, deltas.shape[1] // 4, w,
                                                         C.c_void_p(out.data_ptr()), 1))
        return out

    def nms_grouped(self, boxes, scores, groups, thresh):
        """layers.nms applied independently per group (the box head's per-class loop in one launch pair): kept original
        indices, ascending."""
        if not boxes.is_cuda:
            raise RuntimeError("HipOps.nms_grouped needs CUDA(HIP) tensors; there is no CPU fallback")
        if boxes.shape[0] == 0:
            return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
        order = torch.sort(scores, descending=True, stable=True)[1]
        n = boxes.shape[0]
        if n <= 65536:
            # one SEGMENT PER GROUP (k_nms_mask_seg / k_nms_sweep_seg: one wave per segment, all of them in one launch pair): the groups are independent NMS problems,
            # so sort by (group, score descending) and let every class sweep its own few hundred boxes — a single segment with a group mask swept all candidates
            # block by block on ONE wave (210 us at ~5 000 candidates)
            order = order[torch.sort(groups[order], stable=True)[1]]
            cnts = torch.bincount(groups, minlength=1).to(torch.int32)
            seg_off = (torch.cumsum(cnts, 0) - cnts).to(torch.int32)
            max_n = int(cnts.max().item())
            keep, cnt = self.nms_segments(boxes[order], seg_off, cnts, max_n, thresh)
            flat = (keep.long() + seg_off.long()[:, None])[keep >= 0]
            return torch.sort(order[flat])[0]
        sb = boxes[order].contiguous().float(); sg = groups[order].to(torch.int32).contiguous()
        keep = torch.empty(n, device=boxes.device, dtype=torch.int32); cnt = torch.zeros(1, device=boxes.device, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_nms_grouped(self.ctx.h, C.c_void_p(sb.data_ptr()), None, C.c_void_p(sg.data_ptr()), n, C.c_float(thresh),
                                                      C.c_void_p(keep.data_ptr()), C.c_void_p(cnt.data_ptr()), 1))
        m = int(cnt.item())
        return torch.sort(order[keep[:m].long()])[0]

    def nms(self, boxes, scores, thresh):
        """maskrcnn_benchmark.layers.nms: kept original indices, ascending (sort on the device with torch, sweep in HIP)."""
        if not boxes.is_cuda:
            raise RuntimeError("HipOps.nms needs CUDA(HIP) tensors; there is no CPU fallback")
        if boxes.shape[0] == 0:
            return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
        order = torch.sort(scores, descending=True, stable=True)[1]
        sb = boxes[order].contiguous().float(); n = sb.shape[0]
        keep = torch.empty(max(n, 1), device=boxes.device, dtype=torch.int32); cnt = torch.zeros(1, device=boxes.device, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_nms(self.ctx.h, C.c_void_p(sb.data_ptr()), None, n, C.c_float(thresh), C.c_void_p(keep.data_ptr()), C.c_void_p(cnt.data_ptr()), 1))
        m = int(cnt.item())
        return torch.sort(order[keep[:m].long()])[0]

    # ---- device-only variants used by the no-host-round-trip inference path (nets/maskrcnn.py fast path) ------------------------------------
    def nms_segments(self, boxes, seg_off, seg_n, max_n, thresh, groups=None):
        """layers.nms on len(seg_n) independent segments of `boxes` (each sorted by descending score): (keep [n_seg, max_n] positions relative to the
        segment start, ascending, -1 padded; n_keep [n_seg]) — device tensors, no synchronisation."""
        boxes = boxes.contiguous().float(); n_seg = seg_n.shape[0]
        keep = torch.empty((n_seg, max_n), device=boxes.device, dtype=torch.int32); cnt = torch.empty(n_seg, device=boxes.device, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_nms_segments(self.ctx.h, C.c_void_p(boxes.data_ptr()), C.c_void_p(groups.data_ptr()) if groups is not None else None,
                                                       C.c_void_p(seg_off.data_ptr()), C.c_void_p(seg_n.data_ptr()), n_seg, int(max_n), boxes.shape[0], C.c_float(thresh),
                                                       C.c_void_p(keep.data_ptr()), C.c_void_p(cnt.data_ptr())))
        return keep, cnt

    def roi_align_fpn(self, feats, boxes, level, output_size, scales, sampling_ratio):
        """Pooler.forward in one launch: feats = the 4 FPN maps [1,C,H,W], boxes [n,4], level [n] int32 in 0..3."""
        n = boxes.shape[0]; ph, pw = output_size; Cc = feats[0].shape[1]
        out = torch.empty((n, Cc, ph, pw), device=boxes.device, dtype=torch.float32)
        if n == 0:
            return out
        feats = [f.contiguous().float() for f in feats]; boxes = boxes.contiguous().float(); level = level.to(torch.int32).contiguous()
        fp = (C.c_void_p * 4)(*[f.data_ptr() for f in feats]); Hs = (C.c_int * 4)(*[f.shape[2] for f in feats]); Ws = (C.c_int * 4)(*[f.shape[3] for f in feats])
        sc = (C.c_float * 4)(*[float(x) for x in scales])
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_roi_align_fpn(self.ctx.h, fp, Hs, Ws, sc, Cc, C.c_void_p(boxes.data_ptr()), C.c_void_p(level.data_ptr()), n, ph, pw, sampling_ratio,
                                                        C.c_void_p(out.data_ptr())))
        return out

    # ---- detpost.hip: the detector's selection logic as ordered device kernels -----------------------------------------------------------------------------
    def rpn_select(self, logits, deltas, cell_anchors, strides, K, image_wh):
        """rpn/inference.py:73-105 for all levels in one launch: (boxes [L*K,4], scores [L*K] (-1 = padding), n [L])."""
        L = len(logits); A = logits[0].shape[1]; dev = logits[0].device
        logits = [t.contiguous() for t in logits]; deltas = [t.contiguous() for t in deltas]
        boxes = torch.empty((L * K, 4), device=dev, dtype=torch.float32); scores = torch.empty((L * K,), device=dev, dtype=torch.float32); n = torch.empty((L,), device=dev, dtype=torch.int32)
        lp = (C.c_void_p * L)(*[t.data_ptr() for t in logits]); dp = (C.c_void_p * L)(*[t.data_ptr() for t in deltas])
        hs = (C.c_int * L)(*[t.shape[2] for t in logits]); ws = (C.c_int * L)(*[t.shape[3] for t in logits]); st = (C.c_int * L)(*[int(s) for s in strides])
        key = ("anch", L, A)
        if getattr(self, "_anch_key", None) != key:                 # host copy of the cell anchors (module buffers): made once, outside any capture
            self._anch = (C.c_float * (L * A * 4))(*[float(v) for a in cell_anchors for v in a.detach().cpu().reshape(-1).tolist()]); self._anch_key = key
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_rpn_select(self.ctx.h, L, lp, dp, hs, ws, st, self._anch, A, int(K), int(image_wh[0]), int(image_wh[1]),
                                                     C.c_void_p(boxes.data_ptr()), C.c_void_p(scores.data_ptr()), C.c_void_p(n.data_ptr())))
        return boxes, scores, n

    def rpn_merge(self, boxes, scores, keep, cnt, K, post_nms_top_n, n_final):
        L = cnt.shape[0]; dev = boxes.device
        ob = torch.empty((n_final, 4), device=dev, dtype=torch.float32); osc = torch.empty((n_final,), device=dev, dtype=torch.float32); nv = torch.empty((1,), device=dev, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_rpn_merge(self.ctx.h, C.c_void_p(boxes.data_ptr()), C.c_void_p(scores.data_ptr()), C.c_void_p(keep.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                    L, int(K), int(post_nms_top_n), int(n_final), C.c_void_p(ob.data_ptr()), C.c_void_p(osc.data_ptr()), C.c_void_p(nv.data_ptr())))
        return ob, osc, nv

    def det_class_sort(self, prob, deltas, proposals, objectness, thresh, weights, image_wh):
        prob = prob.contiguous(); deltas = deltas.contiguous(); proposals = proposals.contiguous()
        N, nc = prob.shape; dev = prob.device
        seg = torch.empty(((nc - 1) * N, 4), device=dev, dtype=torch.float32); order = torch.empty((nc - 1, N), device=dev, dtype=torch.int32); seg_n = torch.empty((nc - 1,), device=dev, dtype=torch.int32)
        wv = (C.c_float * 4)(*[float(x) for x in weights])
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_det_class_sort(self.ctx.h, C.c_void_p(prob.data_ptr()), C.c_void_p(deltas.data_ptr()), C.c_void_p(proposals.data_ptr()),
                                                         C.c_void_p(objectness.data_ptr()) if objectness is not None else None, N, nc, C.c_float(thresh), wv, int(image_wh[0]), int(image_wh[1]),
                                                         C.c_void_p(seg.data_ptr()), C.c_void_p(order.data_ptr()), C.c_void_p(seg_n.data_ptr())))
        return seg, order, seg_n

    def det_select(self, prob, seg_boxes, order, keep, cnt, detections_per_img, cap):
        N, nc = prob.shape; dev = prob.device
        ob = torch.empty((cap, 4), device=dev, dtype=torch.float32); osc = torch.empty((cap,), device=dev, dtype=torch.float32); ol = torch.empty((cap,), device=dev, dtype=torch.int64)
        nd = torch.empty((1,), device=dev, dtype=torch.int32); scratch = torch.empty((nc + 3,), device=dev, dtype=torch.int32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_det_select(self.ctx.h, C.c_void_p(prob.data_ptr()), C.c_void_p(seg_boxes.data_ptr()), C.c_void_p(order.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                     C.c_void_p(cnt.data_ptr()), N, nc, int(detections_per_img), int(cap), C.c_void_p(scratch.data_ptr()),
                                                     C.c_void_p(ob.data_ptr()), C.c_void_p(osc.data_ptr()), C.c_void_p(ol.data_ptr()), C.c_void_p(nd.data_ptr())))
        return ob, osc, ol, nd

    def to_nhwc(self, x):
        """[B,C,H,W] f32 -> channels-last copy [B,H,W,C] (vido_nchw_to_nhwc): the layout roi_align_fpn_nhwc reads."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
        x = x.contiguous(); B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float32)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_nchw_to_nhwc(self.ctx.h, C.c_void_p(x.data_ptr()), B, Cc, H, W, C.c_void_p(out.data_ptr())))
        return out

    def roi_align_fpn_nhwc(self, feats_nhwc, boxes, level, output_size, scales, sampling_ratio):
        """Pooler.forward over channels-last maps [1,H,W,C] (to_nhwc of the 4 FPN maps, made once per frame)."""
        n = boxes.shape[0]; ph, pw = output_size; Cc = feats_nhwc[0].shape[3]
        out = torch.empty((n, Cc, ph, pw), device=boxes.device, dtype=torch.float32)
        if n == 0:
            return out
        boxes = boxes.contiguous().float(); level = level.to(torch.int32).contiguous()
        fp = (C.c_void_p * 4)(*[f.data_ptr() for f in feats_nhwc]); Hs = (C.c_int * 4)(*[f.shape[1] for f in feats_nhwc]); Ws = (C.c_int * 4)(*[f.shape[2] for f in feats_nhwc])
        sc = (C.c_float * 4)(*[float(x) for x in scales])
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_roi_align_fpn_nhwc(self.ctx.h, fp, Hs, Ws, sc, Cc, C.c_void_p(boxes.data_ptr()), C.c_void_p(level.data_ptr()), n, ph, pw, sampling_ratio,
                                                             C.c_void_p(out.data_ptr())))
        return out

    def mask_label_image(self, masks, boxes, labels, H, W, thresh=0.5, padding=1):
        """Masker + label image: masks [n,1,M,M], boxes [n,4] (output image), labels [n] int64 -> [H,W] u8."""
        out = torch.empty((H, W), device=masks.device, dtype=torch.uint8)
        n = masks.shape[0]
        masks = masks.contiguous().float(); boxes = boxes.contiguous().float(); labels = labels.contiguous().to(torch.int64)
        self._adopt_stream()
        self.ctx._check(self.ctx.lib.vido_mask_label_image(self.ctx.h, C.c_void_p(masks.data_ptr()) if n else None, C.c_void_p(boxes.data_ptr()) if n else None,
                                                           C.c_void_p(labels.data_ptr()) if n else None, n, masks.shape[-1] if n else 28, padding, C.c_float(thresh), H, W,
                                                           C.c_void_p(out.data_ptr())))
        return out


_C1X1_MIN_TILES = int(os.environ.get("VIDO_CONV1X1_MIN_TILES", "100"))      # (160 through round 5: the fp32-instruction kernel lost to the library at layer4's 112 tiles; the split-bf16 one wins there)


def conv1x1_fills_chip(cout, hw):
    """csrc/conv1x1.hip works on 128 x 128 tiles, one per CU, each walking ALL input channels: a layer with few tiles (layer4's 2048 -> 2048 on 25 x 34: 112; the FPN laterals of
    P4 / P5: 54 / 14) leaves most of the 256 CUs idle for as long as a chip-filling layer takes — measured 119 us against the library's 68 at 112 tiles with the fp32 matrix
    instruction, 64 us with the split-bf16 form (round 6; headline 100.8 -> 103.9 frames/s with layer4 on it, profiles/r6/headline_ab.txt).  Callers keep the library below
    VIDO_CONV1X1_MIN_TILES (default 100)."""
    return (int(cout) // 128) * ((int(hw) + 127) // 128) >= _C1X1_MIN_TILES


def split_bf16x3(x):
    """fp32 tensor -> three bf16 tensors with x0 + x1 + x2 == x EXACTLY (x0 = rne(x), x1 = rne(x - x0), x2 = x - x0 - x1: three 8-bit significands + signs cover the 24 bits
    of fp32; exponents far above the bf16 subnormal range assumed) — the operand form of csrc/conv1x1.hip::k_conv1x1_b3 and of the Winograd kernel's split-bf16 form."""
    x = x.float()
    x0 = x.to(torch.bfloat16); r1 = x - x0.float()
    x1 = r1.to(torch.bfloat16); r2 = r1 - x1.float()
    x2 = r2.to(torch.bfloat16)
    return x0, x1, x2


def split_f16x2(w):
    """fp32 matrix [rows, k] -> (h, l, inv): fp16 planes of the rows scaled by powers of two, w == inv[:, None] * (h + l / 2048) to within 2^-22 |w| (two roundings to 11 bits; 2^-24.5 rms) (h = rne16(s w),
    l = rne16(2048 (s w - h)); s = 1 / inv puts the row's largest |w| into [2^14, 2^15): no overflow; full precision down to 2^-26 of the row's maximum, below that an absolute error under 2^-48 of it) — the weight
    side of csrc/conv1x1.hip::k_conv1x1_b3<.., NP = 2>."""
    w = w.float()
    m = w.abs().amax(1)
    _, ex = torch.frexp(torch.where(m > 0, m, torch.ones_like(m)))          # m = mant 2^ex, mant in [0.5, 1)
    e = torch.where(m > 0, 15 - ex, torch.zeros_like(ex)).clamp(-100, 100).to(torch.int32)
    pow2 = lambda k: ((k + 127) << 23).view(torch.float32)                  # 2^k from its bit pattern: exact on every device (ldexp goes through pow on some)
    ws = w * pow2(e)[:, None]
    h = ws.to(torch.float16)
    l = ((ws - h.float()) * 2048.0).to(torch.float16)
    return h, l, pow2(-e)


def pack_conv1x1(w, layout=0):
    """1x1 convolution weight [cout, cin, 1, 1] (or [cout, cin]) -> the operand order of csrc/conv1x1.hip (layout = HipOps.conv1x1_layout(cin, cout, H * W) = vido_conv1x1_layout):
    0: element (co, k) at [co / 32][k / 8][32 * (k & 1) + co % 32][(k % 8) / 2] (a lane's four operands of a group of four k-pairs are one 16-byte read; a (32-channel block, group)
       is one 1 KB copy piece) — the 128 x 128 tiles on the 32 x 32 x 2 matrix instruction;
    1: at [co / 16][k / 16][16 * (k & 3) + co % 16][(k % 16) / 4] (a lane's 16-byte read = its operands of four k-steps of a 16-row fragment) — the 128 x 112 tiles on 16 x 16 x 4;
    2: the weight split into three bf16 planes (split_bf16x3), plane p of element (co, k) at [co / 32][k / 16][p][32 * ((k % 16) / 8) + co % 32][k % 8] (int16 tensor: a 1 KB copy
       piece = the A operand of v_mfma_f32_32x32x16_bf16 for one (32-channel block, 16 input channels, plane)) — the split-bf16 form k_conv1x1_b3.
    3: the weight as two fp16 planes of its rows scaled by powers of two (split_f16x2), in the order of 2 with two planes, followed by the [cout] inverse row scales
       (flat int16 tensor of 2 cout (cin + 1) elements) — the split-fp16 form, the default.
    None when the kernel does not take the shape."""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if w.dim() == 4 and tuple(w.shape[2:]) != (1, 1):
        return None
    if cout % 128 or cin % 32 or (layout == 1 and cin % 64):
        return None
    if layout == 3:
        h, l, inv = split_f16x2(w.detach().reshape(cout, cin))
        w6 = torch.stack([h, l], 0).view(torch.int16).reshape(2, cout // 32, 32, cin // 16, 2, 8)              # [plane][mb][co32][step][k half][8]
        planes = w6.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1)
        return torch.cat([planes, inv.contiguous().view(torch.int16).reshape(-1)])
    if layout == 2:
        p0, p1, p2 = split_bf16x3(w.detach().reshape(cout, cin).float())
        w6 = torch.stack([p0, p1, p2], 0).view(torch.int16).reshape(3, cout // 32, 32, cin // 16, 2, 8)      # [plane][mb][co32][step][k half][8]
        return w6.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(cout // 32, cin // 16, 3, 64, 8)
    if layout == 1:
        w5 = w.detach().reshape(cout // 16, 16, cin // 16, 4, 4)         # [fragment][co16][group][e][kk]   with k = 16 group + 4 e + kk
        return w5.permute(0, 2, 4, 1, 3).contiguous().reshape(cout // 16, cin // 16, 64, 4)
    w5 = w.detach().reshape(cout // 32, 32, cin // 8, 4, 2)            # [mb][co32][group][kk][half]
    return w5.permute(0, 2, 4, 1, 3).contiguous().reshape(cout // 32, cin // 8, 64, 4)


def pack_conv3x3_h(w):
    """3x3 convolution weight [cout, cin, 3, 3] -> the operand order of csrc/conv3x3h.hip (the direct split-fp16 kernel): two fp16 planes of the output channels scaled by
    powers of two (split_f16x2 over a channel's cin x 9 weights), plane p of element (co, ci, dy, dx) at [co / 32][ci / 16][dy][dx][p][32 * ((ci % 16) / 8) + co % 32][ci % 8],
    followed by the [cout] inverse scales (flat int16 tensor); input channels are padded to a multiple of 16 with zero weights.  None when the kernel does not take the shape."""
    cout, cin0 = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3) or not (cout in (32, 64) or (cout >= 128 and cout % 128 == 0)):
        return None
    cin = (cin0 + 15) // 16 * 16                                              # input channels padded with zero weights
    if cin != cin0:
        w = torch.cat([w.detach(), torch.zeros(cout, cin - cin0, 3, 3, dtype=w.dtype, device=w.device)], 1)
    h, l, inv = split_f16x2(w.detach().reshape(cout, cin * 9))
    planes = torch.stack([h, l], 0).view(torch.int16).reshape(2, cout // 32, 32, cin // 16, 2, 8, 3, 3)        # [plane][mb][co32][chunk][k half][k8][dy][dx]
    planes = planes.permute(1, 3, 6, 7, 0, 4, 2, 5).contiguous().reshape(-1)                                 # [mb][chunk][dy][dx][plane][k half][co32][k8]
    return torch.cat([planes, inv.contiguous().view(torch.int16).reshape(-1)])


def pack_deconv2x2(w):
    """nn.ConvTranspose2d weight [cin, cout, 2, 2] -> the split-fp16 packing of the GEMM matrix [(2 a + b) cout + co][ci] = w[ci][co][a][b] (pack_conv1x1 layout 3): the operand of
    vido_deconv2x2_bias_act."""
    cin, cout = int(w.shape[0]), int(w.shape[1])
    wm = w.detach().permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)
    return pack_conv1x1(wm, 3)


class PackedConv1x1:
    """A 1x1 convolution weight whose operand order depends on the map it meets (pack_conv1x1 layouts 0 / 1): packed lazily per (layout, device), outside any graph capture
    (the warm-up calls of fuse.Graphed come first).  None-like when no layout takes the shape: use PackedConv1x1.make."""

    def __init__(self, w):
        self.w = w.detach(); self.cout = int(w.shape[0]); self.cin = int(w.shape[1]); self._p = {}

    @staticmethod
    def make(w):
        return PackedConv1x1(w) if pack_conv1x1(w, 0) is not None else None

    def get(self, layout, device):
        key = (int(layout), str(device))
        if key not in self._p:
            p = pack_conv1x1(self.w, layout)
            assert p is not None, "conv1x1: layout %d does not take %d -> %d channels" % (layout, self.cin, self.cout)
            self._p[key] = p.to(device)
        return self._p[key]


def pack_conv1x1_skinny(w):
    """1x1 convolution weight [cout, cin, 1, 1] (cin even) -> the operand order of csrc/convsmall.hip::k_conv1x1_skinny: element (co, k) at [co / 32][k / 2][32 * (k & 1) + co % 32],
    output channels padded to a multiple of 32 with zeros."""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    cop = (cout + 31) // 32 * 32
    wz = w.detach().reshape(cout, cin).new_zeros((cop, cin)); wz[:cout] = w.detach().reshape(cout, cin)
    return wz.reshape(cop // 32, 32, cin // 2, 2).permute(0, 2, 3, 1).contiguous().reshape(cop // 32, cin // 2, 64)


def pack_gconv3x3(w, groups):
    """Convolution weight [groups * cpg_out, cpg_in, 3, 3] -> the operand order of csrc/gconv.hip: per (group, 32-channel output block, 8-channel input chunk) [tap][ci][co];
    with 16 or 8 output channels per group: [group][chunk][tap][ci][16 co] (missing output channels zero).  None when the kernels do not take the shape."""
    cout, cpg_in = int(w.shape[0]), int(w.shape[1])
    cpg_out = cout // groups
    if tuple(w.shape[2:]) != (3, 3) or cpg_in % 8 or not (cpg_out % 32 == 0 or cpg_out in (8, 16)):
        return None
    w9 = w.detach().reshape(groups, cpg_out, cpg_in // 8, 8, 9)
    if cpg_out % 32 == 0:
        return w9.reshape(groups, cpg_out // 32, 32, cpg_in // 8, 8, 9).permute(0, 1, 3, 5, 4, 2).contiguous()
    pad = w9.new_zeros((groups, 16, cpg_in // 8, 8, 9)); pad[:, :cpg_out] = w9
    return pad.permute(0, 2, 4, 3, 1).contiguous()


def pack_conv_direct(w):
    """convolution weight [cout, cin, kh, kw] -> the operand order of csrc/convdirect.hip (the library's vido_conv_direct_pack), a flat CPU tensor"""
    import numpy as np
    from ..host import load_library
    lib = load_library()
    lib.vido_conv_direct_packed_floats.restype = C.c_longlong
    cout, cin, kh, kw = (int(v) for v in w.shape)
    wh = np.ascontiguousarray(w.detach().to("cpu", torch.float32).numpy())
    n = int(lib.vido_conv_direct_packed_floats(cin, cout, kh, kw))
    if n <= 0:
        raise RuntimeError("vido_conv_direct_pack: no kernel for %d x %d taps" % (kh, kw))
    out = np.empty(n, np.float32)
    rc = lib.vido_conv_direct_pack(C.c_void_p(wh.ctypes.data), cin, cout, kh, kw, C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError("vido_conv_direct_pack: %d" % rc)
    return torch.from_numpy(out)


def pack_wino3x3(w, form=0):
    """3x3 convolution weight [cout, cin, 3, 3] -> U = G g G^T (float64, rounded once) in the operand order of csrc/wino.hip, as a CPU tensor
    [cout_pad / 32, cin_pad / KC, 16, 64, KC / 2] (the library's vido_wino3x3_pack_form: ONE implementation of the layout, also what a C caller uses).
    form 0: KC by cout (8, or 4 for 32 / 96 .. channels), form 1 (K-split launches): KC = 4."""
    import numpy as np
    from ..host import load_library
    lib = load_library()
    lib.vido_wino3x3_packed_floats_form.restype = C.c_longlong
    cout, cin = int(w.shape[0]), int(w.shape[1])
    assert tuple(w.shape[2:]) == (3, 3) and form in (0, 1, 2)
    wh = np.ascontiguousarray(w.detach().to("cpu", torch.float32).numpy())
    n = int(lib.vido_wino3x3_packed_floats_form(cin, cout, int(form)))
    out = np.empty(n, np.float32)
    rc = lib.vido_wino3x3_pack_form(C.c_void_p(wh.ctypes.data), cin, cout, int(form), C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError("vido_wino3x3_pack_form: %d" % rc)
    kc = 8 if (((cout + 63) // 64) * 64 - cout) < 32 and form == 0 else 4
    cop = ((cout + 63) // 64) * 64 if kc == 8 else ((cout + 31) // 32) * 32
    cip = ((cin + kc - 1) // kc) * kc
    return torch.from_numpy(out).reshape(cop // 32, cip // kc, 16, 64, kc // 2)
