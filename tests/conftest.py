import os, sys
import pytest
try:                      # torch bundles its own HIP runtime: it must be the one the process initialises first,
    import torch          # before libvido_slam_hip.so creates a context (otherwise torch.cuda later finds no GPU)
except ImportError:       # noqa
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def vido():
    import vido_slam_amd
    return vido_slam_amd


class OracleNetOps:
    """box_decode / nms / roi_align for CPU torch tensors through the C oracle — lets the CPU tests run the torch module
    graphs whose product path only accepts the HIP ops (nets.HipOps)."""

    def __init__(self, o):
        self.o = o

    def box_decode(self, deltas, boxes, weights):
        import numpy as np, torch
        if deltas.shape[0] == 0:
            return deltas.clone()
        return torch.from_numpy(self.o.box_decode(deltas.numpy(), boxes.numpy(), np.asarray(weights, np.float32)))

    def nms(self, boxes, scores, thresh):
        import numpy as np, torch
        if boxes.shape[0] == 0:
            return torch.zeros((0,), dtype=torch.int64)
        return torch.from_numpy(self.o.nms(boxes.numpy(), scores.numpy(), float(thresh)).astype(np.int64))

    def nms_grouped(self, boxes, scores, groups, thresh):
        import numpy as np, torch
        keep = []
        g = groups.numpy()
        for c in np.unique(g):
            idx = np.nonzero(g == c)[0]
            keep.append(idx[self.o.nms(boxes.numpy()[idx], scores.numpy()[idx], float(thresh))])
        return torch.from_numpy(np.sort(np.concatenate(keep)).astype(np.int64)) if keep else torch.zeros((0,), dtype=torch.int64)

    def roi_align(self, feat, rois, output_size, spatial_scale, sampling_ratio):
        import torch
        return torch.from_numpy(self.o.roi_align(feat.numpy(), rois.numpy(), float(spatial_scale), output_size[0], output_size[1], sampling_ratio))


@pytest.fixture(scope="session")
def oracle_ops(oracle):
    return OracleNetOps(oracle)
