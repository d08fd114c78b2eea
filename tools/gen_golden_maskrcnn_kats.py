#!/usr/bin/env python3
"""Extracts the known-answer vectors (DATA: inputs + expected outputs) held by the reference's own unit tests
   src/thirdparty/mask_rcnn/src/tests/test_nms.py       (:11-58, :60-217)
   src/thirdparty/mask_rcnn/src/tests/test_box_coder.py (:11-105)
into tests/golden/maskrcnn_kats.npz.  The test modules are executed HERE (build container only) with recording
stubs in place of maskrcnn_benchmark (which cannot be imported: its _C extension does not build), so that the
arrays the tests construct and the expected results they assert are captured verbatim."""
import importlib.util, os, sys, types
import numpy as np
import torch

REF = "/root/reference/src/thirdparty/mask_rcnn/src/tests"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "maskrcnn_kats.npz")
calls, expected = [], []

def box_nms(boxes, scores, thresh):
    calls.append(("nms", boxes.numpy().copy(), scores.numpy().copy(), float(thresh)))
    return np.zeros(0, np.int64)

class BoxCoder:
    def __init__(self, weights): self.weights = weights
    def decode(self, deltas, boxes):
        calls.append(("decode", deltas.numpy().copy(), boxes.numpy().copy(), tuple(self.weights)))
        return torch.zeros(1)

mb = types.ModuleType("maskrcnn_benchmark"); layers = types.ModuleType("maskrcnn_benchmark.layers"); layers.nms = box_nms
modeling = types.ModuleType("maskrcnn_benchmark.modeling"); bc = types.ModuleType("maskrcnn_benchmark.modeling.box_coder"); bc.BoxCoder = BoxCoder
for n, m in (("maskrcnn_benchmark", mb), ("maskrcnn_benchmark.layers", layers), ("maskrcnn_benchmark.modeling", modeling), ("maskrcnn_benchmark.modeling.box_coder", bc)):
    sys.modules[n] = m
np.testing.assert_array_equal = lambda a, b, *k, **kw: expected.append(np.asarray(b).copy())
np.testing.assert_allclose = lambda a, b, *k, **kw: expected.append(np.asarray(b).copy())

for fn in ("test_nms.py", "test_box_coder.py"):
    spec = importlib.util.spec_from_file_location(fn[:-3], os.path.join(REF, fn)); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for name in dir(mod):
        cls = getattr(mod, name)
        if isinstance(cls, type) and name.startswith("Test"):
            obj = cls()
            for meth in sorted(m for m in dir(obj) if m.startswith("test_")):
                getattr(obj, meth)()
assert len(calls) == len(expected), (len(calls), len(expected))
out = {}
k_n = k_d = 0
for c, e in zip(calls, expected):
    if c[0] == "nms":
        out["nms%d_boxes" % k_n], out["nms%d_scores" % k_n], out["nms%d_thresh" % k_n], out["nms%d_keep" % k_n] = c[1], c[2], np.float32(c[3]), e.astype(np.int64); k_n += 1
    else:
        out["dec%d_deltas" % k_d], out["dec%d_boxes" % k_d], out["dec%d_weights" % k_d], out["dec%d_expected" % k_d] = c[1], c[2], np.array(c[3], np.float32), e.astype(np.float32); k_d += 1
np.savez_compressed(OUT, **out)
print("wrote", OUT, "nms cases", k_n, "decode cases", k_d, {k: v.shape for k, v in out.items() if "boxes" in k})
