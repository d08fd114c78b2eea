"""Seeded synthetic inputs for the hot path: no dataset ships with the reference and
there is no network, so bench.py, smoke() and the parity tests all draw from these generators.

make_sequence(): config-2 style stream — textured background translating by a constant integer
flow, rectangles with their own integer velocities (instance labels 1..n_obj), exact flow field,
piecewise-planar metric depth, instance mask.
"""
import numpy as np


def _value_noise(rng, h, w, cell=10):
    gh, gw = h // cell + 2, w // cell + 2
    lat = rng.uniform(0, 255, size=(gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = lat[y0][:, x0]; b = lat[y0][:, x0 + 1]; c = lat[y0 + 1][:, x0]; d = lat[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def make_canvas(h, w, seed=1, n_rect=200):
    """Gray texture: value noise + random axis-aligned rectangles + N(0,2^2) noise, u8."""
    rng = np.random.RandomState(seed)
    img = _value_noise(rng, h, w)
    for _ in range(n_rect):
        rw, rh = rng.randint(8, 60), rng.randint(8, 60)
        x, y = rng.randint(0, max(1, w - rw)), rng.randint(0, max(1, h - rh))
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    img += rng.normal(0, 2.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_frame(w=640, h=480, seed=1):
    """One gray frame (u8, HxW)."""
    return make_canvas(h, w, seed)


def make_batch(n, w=640, h=480, seed=1):
    """n distinct gray frames, shape (n, h, w) u8 (independent seeds)."""
    return np.stack([make_canvas(h, w, seed + 7919 * i) for i in range(n)])


class Sequence:
    """Stream of (gray, bgr, depth_raw, flow, mask) with exact ground truth.

    Background: window sliding over a big canvas by (sx, sy) px / frame  => flow = (-sx, -sy).
    Objects: rectangles, integer velocity (vx, vy) in image space => flow = (vx, vy), label i+1.
    Depth (metric, f32): ground-like ramp 60 m (top) .. 2 m (bottom); objects constant depth.
    `depth_raw` is what the caller hands to TrackRGBD for ChooseData=OMD style (d/factor), i.e.
    metric * depth_map_factor.
    """

    def __init__(self, n_frames=30, w=640, h=480, seed=1, shift=(2, 1), n_obj=5, depth_map_factor=1.0):
        self.n, self.w, self.h = n_frames, w, h
        self.shift = shift
        self.factor = depth_map_factor
        rng = np.random.RandomState(seed + 1000)
        self.canvas = make_canvas(h + shift[1] * n_frames + 8, w + shift[0] * n_frames + 8, seed)
        self.objs = []
        for i in range(n_obj):
            ow, oh = 80, 60
            vx = int(rng.choice([-6, -5, -4, -3, 3, 4, 5, 6])); vy = int(rng.choice([-2, -1, 1, 2]))
            x0 = int(rng.randint(100, w - 100 - ow)); y0 = int(rng.randint(80, h - 80 - oh))
            tex = make_canvas(oh, ow, seed + 31 * (i + 1), n_rect=12)
            self.objs.append(dict(x=x0, y=y0, vx=vx, vy=vy, w=ow, h=oh, tex=tex, depth=8.0 + 3.0 * i))
        ramp = np.linspace(60.0, 2.0, h, dtype=np.float32)[:, None]
        self.bg_depth = np.repeat(ramp, w, axis=1)

    def frame(self, k):
        w, h = self.w, self.h
        ox, oy = self.shift[0] * k, self.shift[1] * k
        gray = self.canvas[oy:oy + h, ox:ox + w].copy()
        flow = np.empty((h, w, 2), np.float32); flow[..., 0] = -self.shift[0]; flow[..., 1] = -self.shift[1]
        depth = self.bg_depth.copy()
        mask = np.zeros((h, w), np.int32)
        for i, o in enumerate(self.objs):
            x, y = o["x"] + o["vx"] * k, o["y"] + o["vy"] * k
            x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + o["w"], w), min(y + o["h"], h)
            if x1 <= x0 or y1 <= y0:
                continue
            gray[y0:y1, x0:x1] = o["tex"][y0 - y:y1 - y, x0 - x:x1 - x]
            flow[y0:y1, x0:x1, 0] = o["vx"]; flow[y0:y1, x0:x1, 1] = o["vy"]
            depth[y0:y1, x0:x1] = o["depth"]
            mask[y0:y1, x0:x1] = i + 1
        bgr = np.stack([gray, np.roll(gray, 1, axis=1), np.roll(gray, 1, axis=0)], axis=-1)
        return gray, np.ascontiguousarray(bgr), depth * np.float32(self.factor), flow, mask

    def __iter__(self):
        for k in range(self.n):
            yield self.frame(k)
