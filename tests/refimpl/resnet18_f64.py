"""MonoDepth2's encoder forward (mono_depth2/src/networks/resnet_encoder.py:87-98 over torchvision's ResNet-18) in float64 numpy, hand-written from the published
architecture — torchvision is absent from this image, so this is the second implementation that pins vido_slam_amd/nets/monodepth2.py::ResnetEncoder18 beyond its
parameter layout.  Works on a state dict with torchvision's names (encoder.conv1.weight, encoder.layer1.0.bn1.running_mean, ...).

  resnet_encoder.py:89     x = (image - 0.45) / 0.225
  :90-92                   features[0] = relu(bn1(conv1(x)))                      conv1: 7x7, stride 2, padding 3, no bias
  :93                      features[1] = layer1(maxpool(features[0]))             maxpool: 3x3, stride 2, padding 1
  :94-96                   features[2..4] = layer2..4
  torchvision BasicBlock:  out = relu(bn1(conv3x3(x, stride))); out = bn2(conv3x3(out)); out = relu(out + (downsample(x) if any else x));
                           downsample = conv1x1(stride) + bn where stride != 1 or the channel count changes; layers = [2, 2, 2, 2] blocks, widths 64 / 128 / 256 / 512,
                           the first block of layer2-4 has stride 2
  BatchNorm2d (eval):      y = (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias
"""
import numpy as np


def conv2d(x, w, stride, pad):
    """x [C, H, W], w [O, C, k, k] -> [O, Ho, Wo]; zero padding, float64."""
    C, H, W = x.shape; O, _, k, _ = w.shape
    xp = np.zeros((C, H + 2 * pad, W + 2 * pad)); xp[:, pad:pad + H, pad:pad + W] = x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = np.zeros((O, Ho, Wo))
    for dy in range(k):
        for dx in range(k):
            patch = xp[:, dy:dy + stride * (Ho - 1) + 1:stride, dx:dx + stride * (Wo - 1) + 1:stride]      # [C, Ho, Wo]
            out += np.tensordot(w[:, :, dy, dx], patch, axes=([1], [0]))
    return out


def bn(x, sd, name):
    g = lambda k: np.asarray(sd[name + "." + k], np.float64)
    return (x - g("running_mean")[:, None, None]) / np.sqrt(g("running_var")[:, None, None] + 1e-5) * g("weight")[:, None, None] + g("bias")[:, None, None]


def maxpool3s2p1(x):
    C, H, W = x.shape
    xp = np.full((C, H + 2, W + 2), -np.inf); xp[:, 1:H + 1, 1:W + 1] = x
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = np.full((C, Ho, Wo), -np.inf)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, xp[:, dy:dy + 2 * (Ho - 1) + 1:2, dx:dx + 2 * (Wo - 1) + 1:2])
    return out


def basic_block(x, sd, name, stride):
    W = lambda k: np.asarray(sd[name + "." + k], np.float64)
    out = np.maximum(bn(conv2d(x, W("conv1.weight"), stride, 1), sd, name + ".bn1"), 0.0)
    out = bn(conv2d(out, W("conv2.weight"), 1, 1), sd, name + ".bn2")
    sc = x
    if (name + ".downsample.0.weight") in sd:
        sc = bn(conv2d(x, W("downsample.0.weight"), stride, 0), sd, name + ".downsample.1")
    return np.maximum(out + sc, 0.0)


def resnet18_encoder_f64(sd, image):
    """sd: state dict (numpy-convertible values) with the names `encoder.*`; image [3, H, W] in [0, 1] -> the five feature maps."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in sd.items()}
    x = (np.asarray(image, np.float64) - 0.45) / 0.225
    x = np.maximum(bn(conv2d(x, np.asarray(sd["encoder.conv1.weight"], np.float64), 2, 3), sd, "encoder.bn1"), 0.0)
    feats = [x]
    x = maxpool3s2p1(x)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = basic_block(x, sd, "encoder.layer%d.0" % li, stride)
        x = basic_block(x, sd, "encoder.layer%d.1" % li, 1)
        feats.append(x)
    return feats
