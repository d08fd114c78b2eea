"""LiteFlowNet forward pass (reference src/thirdparty/flow_net/src/layers.py:39-315 + run_flow_net.py:66-110),
written table-first: every stage is described by a row of LEVELS and built by small factories.  Parameter names
match the reference module tree (netFeatures.netOne.0.weight, netMatching.0.netMain.6.bias, ...)."""
import math
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

LEAK = 0.1
_LFN_NO_WINO = bool(os.environ.get("VIDO_LFN_NO_WINO"))      # experiment: the flow network's dense 3x3 layers on the library's vector-ALU Winograd kernels
#            level: (feature channels, backwarp scale, last-conv kernel, subpixel in-ch, regular. in-ch, dist channels)
LEVELS = {2: (32, 10.0, 7, 130, 131, 49), 3: (64, 5.0, 5, 130, 131, 25), 4: (96, 2.5, 5, 194, 131, 25), 5: (128, 1.25, 3, 258, 131, 9), 6: (192, 0.625, 3, 386, 195, 9)}
MEAN_FIRST = (0.411618, 0.434631, 0.454253)
MEAN_SECOND = (0.410782, 0.433645, 0.452793)


class _Chain(nn.Sequential):
    """nn.Sequential with the reference's index layout (conv, lrelu, conv, ...).  With `epilogue` set (HipOps.bias_act_, GPU only) every
    Conv2d runs without its bias and the bias add + the following LeakyReLU become one in-place HIP pass."""
    epilogue = None
    epilogue_res = None                    # HipOps.bias_res_act_: the chain's last bias add fused with the residual the caller adds (flow + netMain(...), layers.py:160, 199)
    wino = None                            # HipOps: dense 3x3 stride-1 convolutions + bias + LeakyReLU as ONE Winograd launch on the matrix pipe (csrc/wino.hip)

    def forward(self, x, residual=None):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if self.epilogue is not None and isinstance(m, nn.Conv2d) and x.is_cuda and m.bias is not None:
                act = i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)
                if self.wino is not None and not act and i + 1 >= len(mods) and m.out_channels == 2 and hasattr(self.wino, "conv_kxk_c2") and not os.environ.get("VIDO_NO_CONVSMALL"):
                    y = self.wino.conv_kxk_c2(m, x, residual)       # the heads' last layer (32 -> 2, k x k) + bias + `flow +`: one stencil launch (csrc/convsmall.hip)
                    if y is not None:
                        return y
                if self.wino is not None and tuple(m.kernel_size) == (1, 1) and (act or i + 1 >= len(mods)) and hasattr(self.wino, "conv1x1_skinny_conv") and not os.environ.get("VIDO_NO_CONVSMALL"):
                    y = self.wino.conv1x1_skinny_conv(m, x, LEAK if act else 1.0)      # netFeat: few input channels, a large map — one launch, no LDS (csrc/convsmall.hip)
                    if y is not None:
                        x = y; i += 2 if act else 1
                        continue
                if self.wino is not None and (act or i + 1 >= len(mods)) and not _LFN_NO_WINO:
                    y = self.wino.wino3x3_conv(m, x, LEAK if act else 1.0)
                    if y is not None:
                        x = y; i += 2 if act else 1
                        continue
                if self.wino is not None and hasattr(self.wino, "conv_direct_conv") and not os.environ.get("VIDO_NO_CONVDIRECT") and not (residual is not None and i + 1 >= len(mods)):
                    y = self.wino.conv_direct_conv(m, x, LEAK if act else 1.0)        # stem, stride-2 and separable layers: one direct implicit-GEMM launch (csrc/convdirect.hip)
                    if y is not None:
                        x = y; i += 2 if act else 1
                        continue
                x = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
                last = i + (2 if act else 1) >= len(mods)
                if last and residual is not None and self.epilogue_res is not None and not act:
                    x = self.epilogue_res(x.contiguous(), m.bias, residual.contiguous(), 1.0); residual = None
                else:
                    x = self.epilogue(x.contiguous(), m.bias, LEAK if act else 1.0)
                i += 2 if act else 1
            else:
                x = m(x); i += 1
        return x if residual is None else residual + x


def _chain(specs):
    """specs: [(cin, cout, kernel, stride, act)] -> _Chain with the reference's index layout (conv, lrelu, conv, ...)."""
    mods = []
    for cin, cout, k, s, act in specs:
        pad = (k[0] // 2, k[1] // 2) if isinstance(k, tuple) else k // 2
        mods.append(nn.Conv2d(cin, cout, k, s, pad))
        if act:
            mods.append(nn.LeakyReLU(LEAK, inplace=False))
    return _Chain(*mods)


def _flow_head(cin, k):
    return _chain([(cin, 128, 3, 1, True), (128, 64, 3, 1, True), (64, 32, 3, 1, True), (32, 2, k, 1, False)])


def backwarp(x, flow):
    """Bilinear warp of x by flow (pixels), zero padding, align_corners=False grid (layers.py:25-37)."""
    B, _, H, W = flow.shape
    hor = torch.linspace(-1.0 + 1.0 / W, 1.0 - 1.0 / W, W, device=flow.device, dtype=flow.dtype).view(1, 1, 1, W).expand(B, 1, H, W)
    ver = torch.linspace(-1.0 + 1.0 / H, 1.0 - 1.0 / H, H, device=flow.device, dtype=flow.dtype).view(1, 1, H, 1).expand(B, 1, H, W)
    g = torch.cat([hor + flow[:, 0:1] / ((x.shape[3] - 1.0) / 2.0), ver + flow[:, 1:2] / ((x.shape[2] - 1.0) / 2.0)], 1)
    return F.grid_sample(x, g.permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros", align_corners=False)


class _Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.netOne = _chain([(3, 32, 7, 1, True)])
        self.netTwo = _chain([(32, 32, 3, 2, True), (32, 32, 3, 1, True), (32, 32, 3, 1, True)])
        self.netThr = _chain([(32, 64, 3, 2, True), (64, 64, 3, 1, True)])
        self.netFou = _chain([(64, 96, 3, 2, True), (96, 96, 3, 1, True)])
        self.netFiv = _chain([(96, 128, 3, 2, True)])
        self.netSix = _chain([(128, 192, 3, 2, True)])

    def forward(self, x):
        out = []
        for stage in (self.netOne, self.netTwo, self.netThr, self.netFou, self.netFiv, self.netSix):
            x = stage(x); out.append(x)
        return out


class _Matching(nn.Module):
    warp = staticmethod(backwarp)          # LiteFlowNet(warp=HipOps.backwarp) installs the one-pass HIP warp
    def __init__(self, level, corr):
        super().__init__()
        ch, self.scale, k, _, _, _ = LEVELS[level]
        self.corr, self.stride = corr, (2 if level < 4 else 1)
        self.netFeat = _chain([(32, 64, 1, 1, True)]) if level == 2 else nn.Sequential()
        self.netUpflow = None if level == 6 else nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False, groups=2)
        self.netUpcorr = nn.ConvTranspose2d(49, 49, 4, 2, 1, bias=False, groups=49) if level < 4 else None
        self.netMain = _flow_head(49, k)

    fused = None                           # LiteFlowNet(fused=HipOps): the two depthwise transposed convolutions as one-pass HIP kernels (LeakyReLU of the cost volume folded in)

    def forward(self, im1, im2, f1, f2, flow):
        f1, f2 = self.netFeat(f1), self.netFeat(f2)
        fz = self.fused if f1.is_cuda else None
        if flow is not None:
            flow = fz.deconv4s2_depthwise(flow, self.netUpflow.weight) if fz is not None else self.netUpflow(flow)
            f2 = self.warp(f2, flow * self.scale)
        c = self.corr(f1, f2, self.stride)
        if self.netUpcorr is not None:
            c = fz.deconv4s2_depthwise(c, self.netUpcorr.weight, LEAK) if fz is not None else self.netUpcorr(F.leaky_relu(c, LEAK))
        else:
            c = F.leaky_relu(c, LEAK)
        return self.netMain(c, residual=flow) if flow is not None else self.netMain(c)


class _Subpixel(nn.Module):
    warp = staticmethod(backwarp)          # LiteFlowNet(warp=HipOps.backwarp) installs the one-pass HIP warp
    def __init__(self, level):
        super().__init__()
        _, self.scale, k, cin, _, _ = LEVELS[level]
        self.netFeat = _chain([(32, 64, 1, 1, True)]) if level == 2 else nn.Sequential()
        self.netMain = _flow_head(cin, k)

    def forward(self, im1, im2, f1, f2, flow):
        f1, f2 = self.netFeat(f1), self.netFeat(f2)
        f2 = self.warp(f2, flow * self.scale)
        return self.netMain(torch.cat([f1, f2, flow], 1), residual=flow)


class _Regularization(nn.Module):
    warp = staticmethod(backwarp)          # LiteFlowNet(warp=HipOps.backwarp) installs the one-pass HIP warp
    def __init__(self, level):
        super().__init__()
        ch, self.scale, k, _, cin, nd = LEVELS[level]
        self.k = k
        self.netFeat = _chain([(ch, 128, 1, 1, True)]) if level < 5 else nn.Sequential()
        self.netMain = _chain([(cin, 128, 3, 1, True), (128, 128, 3, 1, True), (128, 64, 3, 1, True), (64, 64, 3, 1, True), (64, 32, 3, 1, True), (32, 32, 3, 1, True)])
        self.netDist = _chain([(32, nd, k, 1, False)]) if level >= 5 else _chain([(32, nd, (k, 1), 1, False), (nd, nd, (1, k), 1, False)])
        self.netScaleX = nn.Conv2d(nd, 1, 1); self.netScaleY = nn.Conv2d(nd, 1, 1)

    fused = None                           # LiteFlowNet(fused=HipOps) installs the two-pass HIP form of everything outside the convolutions (lfn_reg_front / lfn_reg_tail)

    def forward(self, im1, im2, f1, f2, flow):
        if self.fused is not None and flow.is_cuda:
            x = self.fused.lfn_reg_front(im1, im2, flow, self.scale, self.netFeat(f1))
            return self.fused.lfn_reg_tail(self.netDist(self.netMain(x)), flow, self.netScaleX, self.netScaleY, self.k)
        diff = (im1 - self.warp(im2, flow * self.scale)).pow(2.0).sum(1, True).sqrt()
        centred = flow - flow.flatten(2).mean(2, True).unsqueeze(-1)
        d = self.netDist(self.netMain(torch.cat([diff, centred, self.netFeat(f1)], 1))).pow(2.0).neg()
        d = (d - d.max(1, True)[0]).exp()
        div = d.sum(1, True).reciprocal()
        pad = (self.k - 1) // 2
        sx = self.netScaleX(d * F.unfold(flow[:, 0:1], self.k, 1, pad).view_as(d)) * div
        sy = self.netScaleY(d * F.unfold(flow[:, 1:2], self.k, 1, pad).view_as(d)) * div
        return torch.cat([sx, sy], 1)


class LiteFlowNet(nn.Module):
    """`correlation`: callable (first, second, stride) -> cost volume.  On the GPU pass HipOps(ctx).correlation (the HIP
    kernel); the CPU tests pass correlation_torch_reference."""

    def __init__(self, correlation, epilogue=None, warp=None, fused=None, pair_batch=True):
        super().__init__()
        self.pair_batch = pair_batch
        self.netFeatures = _Features()
        self.netMatching = nn.ModuleList([_Matching(l, correlation) for l in (2, 3, 4, 5, 6)])
        self.netSubpixel = nn.ModuleList([_Subpixel(l) for l in (2, 3, 4, 5, 6)])
        self.netRegularization = nn.ModuleList([_Regularization(l) for l in (2, 3, 4, 5, 6)])
        if epilogue is not None:
            res = getattr(getattr(epilogue, "__self__", None), "bias_res_act_", None)      # the residual form of the same HipOps object
            hip = getattr(epilogue, "__self__", None)
            wino = hip if hasattr(hip, "wino3x3_conv") and not os.environ.get("VIDO_NO_WINO") else None
            for m in self.modules():
                if isinstance(m, _Chain):
                    m.epilogue = epilogue; m.epilogue_res = res; m.wino = wino
        if warp is not None:
            for m in self.modules():
                if isinstance(m, (_Matching, _Subpixel, _Regularization)):
                    m.warp = warp
        if fused is not None:
            for m in self.modules():
                if isinstance(m, (_Regularization, _Matching)):
                    m.fused = fused
        # per-channel means as (non-persistent) buffers: the reference builds them with new_tensor inside forward (layers.py:286-287), a host-to-device
        # copy per call that a hipGraph capture cannot contain; not part of the state dict, so the reference's checkpoints still load unchanged
        self.register_buffer("_mean_first", torch.tensor(MEAN_FIRST).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("_mean_second", torch.tensor(MEAN_SECOND).view(1, 3, 1, 1), persistent=False)

    @torch.no_grad()
    def forward(self, first, second):
        first = first - self._mean_first
        second = second - self._mean_second
        if self.pair_batch and first.is_cuda:
            # the two images as one batch of 2 through the shared-weight feature pyramid and the image pyramid (same arithmetic per image; half the launches)
            fb = self.netFeatures(torch.cat([first, second], 0)); B = first.shape[0]
            f1, f2 = [t[:B] for t in fb], [t[B:] for t in fb]
            pb = [torch.cat([first, second], 0)]
            for l in range(1, 6):
                pb.append(F.interpolate(pb[-1], size=fb[l].shape[2:], mode="bilinear", align_corners=False))
            p1, p2 = [t[:B] for t in pb], [t[B:] for t in pb]
        else:
            f1, f2 = self.netFeatures(first), self.netFeatures(second)
            p1, p2 = [first], [second]
            for l in range(1, 6):
                size = f1[l].shape[2:]
                p1.append(F.interpolate(p1[-1], size=size, mode="bilinear", align_corners=False))
                p2.append(F.interpolate(p2[-1], size=size, mode="bilinear", align_corners=False))
        flow = None
        for l in (-1, -2, -3, -4, -5):          # level 6 -> 2
            flow = self.netMatching[l](p1[l], p2[l], f1[l], f2[l], flow)
            flow = self.netSubpixel[l](p1[l], p2[l], f1[l], f2[l], flow)
            flow = self.netRegularization[l](p1[l], p2[l], f1[l], f2[l], flow)
        return flow * 20.0


@torch.no_grad()
def analyse_flow(net, previous_bgr, current_bgr):
    """run_flow_net.py:66-110: HxWx3 u8 BGR pair -> HxWx2 f32 flow (BGR->RGB, /255, bilinear resize to x32, forward,
    bilinear resize back, rescale u by W/W', v by H/H')."""
    dev = next(net.parameters()).device
    def prep(img):
        t = (img.to(dev).flip(-1) if torch.is_tensor(img) else torch.as_tensor(img[:, :, ::-1].copy(), device=dev)).permute(2, 0, 1).float().div(255.0).unsqueeze(0)   # a device tensor stays on the device
        return t
    a, b = prep(previous_bgr), prep(current_bgr)
    H, W = a.shape[2], a.shape[3]
    Hp, Wp = int(math.floor(math.ceil(H / 32.0) * 32.0)), int(math.floor(math.ceil(W / 32.0) * 32.0))
    a = F.interpolate(a, size=(Hp, Wp), mode="bilinear", align_corners=False)
    b = F.interpolate(b, size=(Hp, Wp), mode="bilinear", align_corners=False)
    flow = F.interpolate(net(a, b), size=(H, W), mode="bilinear", align_corners=False)
    flow[:, 0] *= float(W) / float(Wp); flow[:, 1] *= float(H) / float(Hp)
    return flow[0].permute(1, 2, 0).contiguous()
