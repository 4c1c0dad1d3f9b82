"""Five lattice builds of the bench batch (32 images at 321 x 321) for profiles/gbuild_timeline.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
from wsscam import _lib, synth  # noqa: E402

ctx = _lib.Context(0)
_, rgb, _ = synth.image_batch(32, 321, 0)
rgb_dev = ctx.to_device(rgb)
for i in range(5):
    ctx.sync()
    t0 = time.perf_counter()
    c = _lib.Crf(ctx, rgb_dev, 32, 321, 321, 1.5, 40.0, 13.0)
    ctx.sync()
    print("build %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
    c.close()
