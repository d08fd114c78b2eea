"""X-101-32x8d grouped 3x3 convolutions (groups = 32) at the detector's 800x1088 feed: MIOpen's time per stage shape"""
import os, sys, time
import torch, torch.nn.functional as F
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
tot = 0
for name, ch, h, w, stride, count in (("res2", 256, 200, 272, 1, 3), ("res3a", 512, 200, 272, 2, 1), ("res3", 512, 100, 136, 1, 3), ("res4a", 1024, 100, 136, 2, 1), ("res4", 1024, 50, 68, 1, 22),
                                       ("res5a", 2048, 50, 68, 2, 1), ("res5", 2048, 25, 34, 1, 2), ("fpn3x3", 256, 200, 272, 1, 1), ("fpn3x3_p3", 256, 100, 136, 1, 1)):
    g = 1 if name.startswith("fpn") else 32
    x = torch.randn(1, ch, h, w, device="cuda"); wt = torch.randn(ch, ch // g, 3, 3, device="cuda")
    t = timeit(lambda: F.conv2d(x, wt, None, stride, 1, 1, g))
    oh, ow = h // stride, w // stride
    gf = 2.0 * ch * (ch // g) * 9 * oh * ow / 1e9
    mb = (x.numel() + ch * oh * ow) * 4 / 1e6
    print("%-10s C %4d %3dx%3d s%d g%2d: %7.1f us  %6.2f GF (%5.1f TF/s)  %6.1f MB (%.2f TB/s)  x%d" % (name, ch, h, w, stride, g, t, gf, gf / t * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e3, mb, mb / t, count))
    tot += t * count
print("grouped/FPN 3x3 total per frame: %.2f ms" % (tot / 1e3))
