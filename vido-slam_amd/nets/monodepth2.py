"""MonoDepth2 forward pass (reference src/thirdparty/mono_depth2/src/networks/{resnet_encoder,depth_decoder}.py,
layers.py:106-137,196-199, run_mono_depth.py:101-156).  The reference encoder is torchvision's ResNet-18 (absent from
this image): it is restated here with the same parameter names (encoder.conv1.weight, encoder.layer1.0.bn1...), so
the published encoder.pth / depth.pth load unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Basic(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    _ep = None                             # nets/fuse.py::fold_batchnorm

    def forward(self, x):
        if self._ep is not None and x.is_cuda:
            ep = self._ep
            y = ep(F.conv2d(x, self._w1, None, self.conv1.stride, 1), self._b1, None, 0.0)
            sc = x if self.downsample is None else ep(F.conv2d(x, self._wd, None, self.downsample[0].stride), self._bd, None, 1.0)
            return ep(F.conv2d(y, self._w2, None, 1, 1), self._b2, sc.contiguous(), 0.0)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ResNet18Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False); self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(_Basic(64, 64, 1), _Basic(64, 64, 1))
        self.layer2 = nn.Sequential(_Basic(64, 128, 2), _Basic(128, 128, 1))
        self.layer3 = nn.Sequential(_Basic(128, 256, 2), _Basic(256, 256, 1))
        self.layer4 = nn.Sequential(_Basic(256, 512, 2), _Basic(512, 512, 1))
        self.fc = nn.Linear(512, 1000)            # present in the checkpoint, unused by the depth path


class ResnetEncoder18(nn.Module):
    num_ch_enc = (64, 64, 128, 256, 512)

    def __init__(self):
        super().__init__()
        self.encoder = _ResNet18Trunk()

    _ep = None

    def forward(self, image):                     # resnet_encoder.py:87-98
        e = self.encoder
        if self._ep is not None and image.is_cuda:
            x = self._ep(F.conv2d((image - 0.45) / 0.225, self._w1, None, 2, 3), self._b1, None, 0.0)
        else:
            x = F.relu(e.bn1(e.conv1((image - 0.45) / 0.225)))
        feats = [x]
        x = e.layer1(F.max_pool2d(x, 3, 2, 1)); feats.append(x)
        for layer in (e.layer2, e.layer3, e.layer4):
            x = layer(x); feats.append(x)
        return feats


class _Conv3x3(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1); self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv3x3(cin, cout); self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    """depth_decoder.py:18-66; `decoder` ModuleList order = (upconv 4,0), (4,1), (3,0) ... (0,1), dispconv 0..3."""
    num_ch_dec = (16, 32, 64, 128, 256)

    def __init__(self, num_ch_enc=ResnetEncoder18.num_ch_enc, scales=range(4)):
        super().__init__()
        self.scales = list(scales); mods = []; self.index = {}
        for i in range(4, -1, -1):
            cin = num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.index[("upconv", i, 0)] = len(mods); mods.append(_ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (num_ch_enc[i - 1] if i > 0 else 0)
            self.index[("upconv", i, 1)] = len(mods); mods.append(_ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            self.index[("dispconv", s)] = len(mods); mods.append(_Conv3x3(self.num_ch_dec[s], 1))
        self.decoder = nn.ModuleList(mods)

    _ops = None                            # nets/fuse.py::fold_batchnorm: HipOps for the one-pass decoder glue

    def forward_fused(self, feats):
        """("disp", 0) only (the node uses nothing else, run_mono_depth.py:137), with the glue of every level as single passes: bias + ELU (vido_bias_unary), upsample + cat +
        the next block's reflection pad (vido_upcat_reflect), bias + sigmoid; the convolutions stay the library's."""
        ops = self._ops; x = feats[-1]
        for i in range(4, -1, -1):
            b0 = self.decoder[self.index[("upconv", i, 0)]].conv; b1 = self.decoder[self.index[("upconv", i, 1)]].conv
            x = ops.bias_unary_(F.conv2d(b0.pad(x), b0.conv.weight, None), b0.conv.bias, "elu")
            x = ops.upcat_reflect(x, feats[i - 1].contiguous() if i > 0 else None)
            x = ops.bias_unary_(F.conv2d(x, b1.conv.weight, None), b1.conv.bias, "elu")
        d0 = self.decoder[self.index[("dispconv", 0)]]
        return {("disp", 0): ops.bias_unary_(F.conv2d(d0.pad(x), d0.conv.weight, None), d0.conv.bias, "sigmoid")}

    def forward(self, feats):
        if self._ops is not None and feats[-1].is_cuda and feats[-1].shape[0] == 1:
            return self.forward_fused(feats)
        out = {}; x = feats[-1]
        for i in range(4, -1, -1):
            x = self.decoder[self.index[("upconv", i, 0)]](x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if i > 0:
                x = torch.cat([x, feats[i - 1]], 1)
            x = self.decoder[self.index[("upconv", i, 1)]](x)
            if i in self.scales:
                out[("disp", i)] = torch.sigmoid(self.decoder[self.index[("dispconv", i)]](x))
        return out


class MonoDepth2(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = ResnetEncoder18(); self.depth_decoder = DepthDecoder()

    @torch.no_grad()
    def forward(self, image):
        return self.depth_decoder(self.encoder(image))[("disp", 0)]


@torch.no_grad()
def analyse_depth(net, bgr, feed=(192, 640), ops=None):
    """run_mono_depth.py:101-156: HxWx3 u8 BGR -> HxW u16 (area-resize to 640x192, BGR->RGB, /255, forward, bilinear resize of
    disp_0 back, min-max normalise to [0, 65536])."""
    dev = next(net.parameters()).device
    if ops is not None and torch.is_tensor(bgr) and bgr.is_cuda and bgr.dtype == torch.uint8:
        H, W = bgr.shape[0], bgr.shape[1]
        x = ops.area_feed(bgr.contiguous(), feed, 255.0)             # flip + float + area resize + / 255 in one HIP pass (vido_area_feed), identical values
    else:
        t = (bgr.to(dev).flip(-1) if torch.is_tensor(bgr) else torch.as_tensor(bgr[:, :, ::-1].copy(), device=dev)).permute(2, 0, 1).float().unsqueeze(0)
        H, W = t.shape[2], t.shape[3]
        x = F.interpolate(t, size=feed, mode="area").div(255.0)         # cv2.INTER_AREA
    disp = F.interpolate(net(x), size=(H, W), mode="bilinear", align_corners=False)[0, 0]
    if ops is not None and disp.is_cuda and hasattr(ops, "minmax_norm_u16"):
        return ops.minmax_norm_u16(disp)
    lo, hi = disp.min(), disp.max()
    return ((disp - lo) / (hi - lo + 1e-12) * 65536.0).clamp(0, 65535).to(torch.int32)
