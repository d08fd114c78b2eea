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


class Scene3D:
    """Geometrically consistent RGB-D + flow + mask sequence: a textured ground plane
    (y = ground_y, camera looks along +z with y down) and a far wall (z = wall_z), plus moving fronto-parallel
    textured squares (instance labels 1..), seen from a camera that drives forward with a slow yaw.  Everything is
    rendered by ray casting, so depth is exact, flow(k) is the exact projection of the same surface point into
    frame k+1 (objects move between the frames), and the ground-truth poses are known."""

    def __init__(self, n_frames=12, w=640, h=480, K=(500.0, 500.0, 319.5, 239.5), seed=3, step=0.25, yaw_deg=0.4,
                 objects=((-2.0, 0.2, 9.0, 0.10, 0.0, 0.05),), obj_half=1.1, wall_z=32.0):
        self.n, self.w, self.h, self.K = n_frames, w, h, K
        self.ground_y, self.wall_z = 1.6, wall_z
        self.tex = make_canvas(1024, 1024, seed, n_rect=600).astype(np.float32)
        self.otex = [make_canvas(256, 256, seed + 17 * (i + 1), n_rect=40).astype(np.float32) for i in range(len(objects))]
        self.objects = objects          # (x, y, z, vx, vy, vz) of the square centre at frame 0, metres / frame
        self.obj_half = obj_half
        self.poses = []                 # camera-to-world 4x4
        T = np.eye(4)
        for k in range(n_frames + 1):
            self.poses.append(T.copy())
            a = np.deg2rad(yaw_deg)
            d = np.eye(4); d[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]); d[2, 3] = step
            T = T @ d
        fx, fy, cx, cy = K
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        self.rays = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)        # camera frame, z = 1
        self.uv = np.stack([u, v], -1)

    def _sample(self, tex, a, b, scale):
        th, tw = tex.shape
        x = (a * scale) % (tw - 1); y = (b * scale) % (th - 1)
        x0 = np.floor(x).astype(int); y0 = np.floor(y).astype(int); fx = x - x0; fy = y - y0
        return tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) + tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy

    def Tcw(self, k):
        return np.linalg.inv(self.poses[k])

    def frame(self, k):
        w, h = self.w, self.h; fx, fy, cx, cy = self.K
        Twc, Tcw_next = self.poses[k], np.linalg.inv(self.poses[k + 1])
        o = Twc[:3, 3]; d = self.rays @ Twc[:3, :3].T                                   # world ray directions (per unit camera z)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[..., 1] > 1e-6, (self.ground_y - o[1]) / d[..., 1], np.inf)
            tw_ = np.where(d[..., 2] > 1e-6, (self.wall_z - o[2]) / d[..., 2], np.inf)
        t = np.minimum(tg, tw_); is_ground = tg <= tw_
        X = o + d * t[..., None]
        gray = np.where(is_ground, self._sample(self.tex, X[..., 0] + 200, X[..., 2] + 50, 14.0), self._sample(self.tex, X[..., 0] + 300, X[..., 1] + 40, 9.0))
        Xn = X.copy()                                                                  # where the surface point is at frame k+1 (static)
        mask = np.zeros((h, w), np.int32)
        for i, (ox, oy, oz, vx, vy, vz) in enumerate(self.objects):
            c = np.array([ox + vx * k, oy + vy * k, oz + vz * k])
            with np.errstate(divide="ignore", invalid="ignore"):
                to = np.where(d[..., 2] > 1e-6, (c[2] - o[2]) / d[..., 2], np.inf)
            P = o + d * to[..., None]
            hit = (np.abs(P[..., 0] - c[0]) < self.obj_half) & (np.abs(P[..., 1] - c[1]) < self.obj_half) & (to < t) & (to > 0)
            gray = np.where(hit, self._sample(self.otex[i], P[..., 0] - c[0] + 2, P[..., 1] - c[1] + 2, 50.0), gray)
            t = np.where(hit, to, t); X = np.where(hit[..., None], P, X); Xn = np.where(hit[..., None], P + np.array([vx, vy, vz]), Xn)
            mask[hit] = i + 1
        depth = t.astype(np.float32)                                                   # rays have camera z = 1  =>  t is the depth
        Xc = Xn @ Tcw_next[:3, :3].T + Tcw_next[:3, 3]
        un = Xc[..., 0] / Xc[..., 2] * fx + cx; vn = Xc[..., 1] / Xc[..., 2] * fy + cy
        flow = np.stack([un - self.uv[..., 0], vn - self.uv[..., 1]], -1).astype(np.float32)
        g8 = np.clip(np.rint(gray), 0, 255).astype(np.uint8)
        return g8, depth, flow, mask


def convoy_scene(n_frames, w=640, h=480, seed=5, step=0.25):
    """BASELINE configs[1]-[3] chained: a 640x480 Scene3D with FIVE dynamic objects that drive ahead of the camera at roughly its own
    speed (so they stay in view for the whole clip) with distinct lateral / forward velocities — scene flow 0.26-0.34 m per frame, above SFMgThres —
    each about 110 px wide (>= 150 dense samples, depth < ThDepthOBJ)."""
    objs = ((-4.4, 0.55, 8.0, -0.004, 0.0, step + 0.010), (-2.2, 0.55, 9.5, -0.006, 0.0, step + 0.020), (0.0, 0.55, 11.0, 0.004, 0.0, step + 0.000),
            (2.2, 0.55, 9.5, 0.006, 0.0, step + 0.015), (4.4, 0.55, 8.0, 0.004, 0.0, step + 0.005))
    # the far wall stays ahead of the camera for the whole clip (48 m for clips of up to 128 frames — the default bench — and 16 m beyond the end of longer ones: a 200-step run used
    # to drive through it at frame 192 and finish on a scene without structure)
    return Scene3D(n_frames=n_frames, w=w, h=h, seed=seed, step=step, yaw_deg=0.05, objects=objs, obj_half=0.9, wall_z=max(48.0, n_frames * step + 16.0))


def gray_to_bgr(gray):
    """HxW u8 -> HxWx3 u8 with B = G = R: cvtColor(BGR2GRAY) of it is the gray image again ((1868 + 9617 + 4899) g + 8192) >> 14 == g."""
    import numpy as _np
    return _np.ascontiguousarray(_np.repeat(gray[:, :, None], 3, axis=2))
