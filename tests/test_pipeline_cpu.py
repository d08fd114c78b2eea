"""Host-side pieces of the in-process hand-off that need no GPU: the fixed-point gray conversion equals the facade's (Tracking.cc:327-340)."""
import numpy as np
import torch
from vido_slam_amd import pipeline


def test_bgr_to_gray_is_the_14_bit_fixed_point_formula():
    rng = np.random.RandomState(0)
    bgr = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    bgr[0, 0] = (255, 255, 255); bgr[0, 1] = (0, 0, 0); bgr[0, 2] = (255, 0, 0); bgr[0, 3] = (0, 255, 0); bgr[0, 4] = (0, 0, 255)
    got = pipeline.bgr_to_gray(torch.from_numpy(bgr)).numpy()
    b, g, r = (bgr[..., c].astype(np.int64) for c in range(3))
    ref = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)
    assert got[0, 0] == 255 and got[0, 1] == 0                      # the three weights sum to 2^14
