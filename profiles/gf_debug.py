"""debug: GF (Gaussian blur inside the update kernel) vs the unfused path at several sizes; prints the mismatch count"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "wsss-analysis_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from wsscam import _lib
from tests import helpers
from tests.test_gpu_crf import _gpu_crf
ctx = _lib.Context(0)
for (H, W, M, B) in [(500, 375, 5, 1), (640, 640, 5, 1), (1088, 1088, 5, 1), (321, 321, 21, 16)]:
    rng = np.random.default_rng(1)
    cases = [helpers.synth_crf_case(rng, H, W, M) for _ in range(B)]
    cfg = (1.5, 3, 40, 13, 10, 4)
    os.environ["WSC_CRF_NO_GFUSE"] = "1"
    q0, a0, vg, vb = _gpu_crf(ctx, [c[0] for c in cases], [c[1] for c in cases], cfg)
    os.environ["WSC_CRF_NO_GFUSE"] = "0"
    q1, a1, _, _ = _gpu_crf(ctx, [c[0] for c in cases], [c[1] for c in cases], cfg)
    print(H, W, M, B, "on_chip", _gpu_crf.on_chip, "max", np.abs(q0 - q1).max(), "nbad", int((q0 != q1).sum()), flush=True)
