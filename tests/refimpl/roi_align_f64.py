"""ROI-Align forward in float64 numpy, written from the rule of maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:15-122 (sampling points and bilinear weights) and :125-217
(the average over the roi_bin_grid_h x roi_bin_grid_w samples of a bin).  Vectorised over the sample grid of one ROI — deliberately NOT the loop structure of
oracle/nets_oracle.c, so that the two can check each other.

Rule, per ROI (batch index, x1, y1, x2, y2) — ROIAlign_cpu.cpp:
  :147-150  roi_start/end = coordinate * spatial_scale                      (no rounding)
  :157-160  roi_width / roi_height = max(end - start, 1); bin_size = roi_size / pooled_size
  :163-167  grid = sampling_ratio if > 0 else ceil(roi_size / pooled_size)
  :37-43    sample (ph, iy): y = roi_start_h + ph * bin_size_h + (iy + 0.5) * bin_size_h / grid_h   (x alike)
  :47-62    a sample with y < -1 or y > height or x < -1 or x > width contributes 0
  :64-69    y = max(y, 0); x = max(x, 0)
  :71-88    y_low = int(y); if y_low >= height - 1: y_high = y_low = height - 1, y = y_low  else y_high = y_low + 1   (x alike)
  :90-93    ly = y - y_low, lx = x - x_low, hy = 1 - ly, hx = 1 - lx; weights hy hx, hy lx, ly hx, ly lx on (low, low), (low, high), (high, low), (high, high)
  :197-211  bin value = sum of the weighted samples / (grid_h * grid_w)
"""
import numpy as np


def roi_align_f64(feat, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio):
    """feat [B, C, H, W], rois [R, 5] (batch index, x1, y1, x2, y2) -> [R, C, pooled_h, pooled_w] float64."""
    feat = np.asarray(feat, np.float64); rois = np.asarray(rois, np.float64)
    B, C, H, W = feat.shape
    out = np.zeros((len(rois), C, pooled_h, pooled_w), np.float64)
    for r, roi in enumerate(rois):
        b = int(roi[0])
        sw, sh, ew, eh = (roi[1:5] * spatial_scale).tolist()
        rw, rh = max(ew - sw, 1.0), max(eh - sh, 1.0)
        bh, bw = rh / pooled_h, rw / pooled_w
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / pooled_h))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pooled_w))
        # all sample coordinates of the ROI: ys [pooled_h * gh], xs [pooled_w * gw]
        ys = sh + np.repeat(np.arange(pooled_h), gh) * bh + (np.tile(np.arange(gh), pooled_h) + 0.5) * bh / gh
        xs = sw + np.repeat(np.arange(pooled_w), gw) * bw + (np.tile(np.arange(gw), pooled_w) + 0.5) * bw / gw
        vy = ~((ys < -1.0) | (ys > H)); vx = ~((xs < -1.0) | (xs > W))

        def axis(v, n):
            v = np.maximum(v, 0.0)
            lo = v.astype(np.int64)                         # truncation of a non-negative number
            top = lo >= n - 1
            lo = np.where(top, n - 1, lo); hi = np.where(top, n - 1, lo + 1)
            v = np.where(top, lo.astype(np.float64), v)
            l = v - lo
            return lo, hi, 1.0 - l, l

        yl, yh, hy, ly = axis(ys, H); xl, xh, hx, lx = axis(xs, W)
        hy = hy * vy; ly = ly * vy; hx = hx * vx; lx = lx * vx    # an invalid sample: all four weights 0
        f = feat[b]                                               # [C, H, W]
        # separable bilinear gather: rows first, then columns
        rows = f[:, yl, :] * hy[None, :, None] + f[:, yh, :] * ly[None, :, None]                   # [C, S_y, W]
        samp = rows[:, :, xl] * hx[None, None, :] + rows[:, :, xh] * lx[None, None, :]             # [C, S_y, S_x]
        out[r] = samp.reshape(C, pooled_h, gh, pooled_w, gw).sum(axis=(2, 4)) / (gh * gw)
    return out
