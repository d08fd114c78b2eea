"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does (tests/test_layout.py greps for that)."""
import ctypes as C, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])

def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvido_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB

class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int)]
KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])

class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int), ("n_levels", C.c_int), ("ini_th", C.c_int), ("min_th", C.c_int),
                ("scale_factor", C.c_float), ("scale", C.c_float * 16), ("inv_scale", C.c_float * 16),
                ("n_per_level", C.c_int * 16), ("umax", C.c_int * 16)]

def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)

def orb_params(n_features=2000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7):
    p = OrbParams()
    lib().vo_orb_params_init(C.byref(p), n_features, C.c_float(scale_factor), n_levels, ini_th, min_th)
    return p

def level_size(p, w, h, l):
    lw, lh = C.c_int(), C.c_int()
    lib().vo_level_size(C.byref(p), w, h, l, C.byref(lw), C.byref(lh))
    return lw.value, lh.value

def bgr2gray(img, rgb_order=False):
    img = np.ascontiguousarray(img, np.uint8); h, w, c = img.shape
    out = np.empty((h, w), np.uint8)
    lib().vo_bgr2gray(_p(img), w * c, w, h, c, int(rgb_order), _p(out), w)
    return out

def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    out = np.empty((dh, dw), np.uint8)
    lib().vo_resize_linear_u8(_p(src), w, w, h, _p(out), dw, dw, dh)
    return out

def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    out = np.empty_like(src)
    lib().vo_gaussian_blur7(_p(src), w, w, h, _p(out), w)
    return out

def fast_atan2(y, x):
    f = lib().vo_fast_atan2; f.restype = C.c_float
    return f(C.c_float(y), C.c_float(x))

def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    out = np.empty((w * h, 3), np.int32)
    n = lib().vo_fast9_16(_p(img), w, w, h, threshold, int(nonmax), _p(out), w * h)
    return out[:n].copy()

def fast_score_map(img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().vo_fast_score_map(_p(img), w, w, h, _p(out), w)
    return out

def level_candidates(p, img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    cap = w * h // 4 + 16
    cx = np.empty(cap, np.float32); cy = np.empty(cap, np.float32); cr = np.empty(cap, np.float32)
    n = lib().vo_level_candidates(C.byref(p), _p(img), w, w, h, _p(cx), _p(cy), _p(cr), cap)
    return cx[:n].copy(), cy[:n].copy(), cr[:n].copy()

def distribute_octree(cx, cy, cr, minX, maxX, minY, maxY, N):
    cx = np.ascontiguousarray(cx, np.float32); cy = np.ascontiguousarray(cy, np.float32); cr = np.ascontiguousarray(cr, np.float32)
    out = np.empty(max(len(cx), 1), np.int32)
    n = lib().vo_distribute_octree(_p(cx), _p(cy), _p(cr), len(cx), minX, maxX, minY, maxY, N, _p(out), len(out))
    return out[:n].copy()

def ic_angle(img, x, y, p):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    f = lib().vo_ic_angle; f.restype = C.c_float
    return f(_p(img), w, int(x), int(y), p.umax)

def brief(blurred, x, y, angle):
    img = np.ascontiguousarray(blurred, np.uint8); h, w = img.shape
    d = np.empty(32, np.uint8)
    lib().vo_brief(_p(img), w, int(x), int(y), C.c_float(angle), _p(d))
    return d

def orb_pyramid(p, gray):
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    offs = (C.c_int * 16)()
    total = lib().vo_orb_pyramid(C.byref(p), _p(gray), w, w, h, None, offs)
    buf = np.empty(total, np.uint8)
    lib().vo_orb_pyramid(C.byref(p), _p(gray), w, w, h, _p(buf), offs)
    levels = []
    for l in range(p.n_levels):
        lw, lh = level_size(p, w, h, l)
        levels.append(buf[offs[l]:offs[l] + lw * lh].reshape(lh, lw))
    return levels

def orb_extract(p, gray, cap=None):
    gray = np.ascontiguousarray(gray, np.uint8); h, w = gray.shape
    cap = cap or (p.n_features * 2 + 64)
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); ncand = (C.c_int * 16)()
    n = lib().vo_orb_extract(C.byref(p), _p(gray), w, w, h, _p(kps), _p(desc), cap, ncand)
    assert n <= cap
    return kps[:n].copy(), desc[:n].copy(), list(ncand)[:p.n_levels]

def hamming_match(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    idx = np.empty(len(a), np.int32); dist = np.empty(len(a), np.int32)
    lib().vo_hamming_match(_p(a), len(a), _p(b), len(b), _p(idx), _p(dist))
    return idx, dist
