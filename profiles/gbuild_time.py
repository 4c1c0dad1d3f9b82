import sys, time, os
import numpy as np
sys.path.insert(0, "wsss-analysis_amd")
from wsscam import _lib, synth
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
for (H, W) in [(321, 321), (375, 500), (333, 500), (500, 500)]:
    img = np.stack([synth.synth_image(rng, H, W)])
    rgb = ctx.to_device(img)
    ctx.sync()
    t0 = time.perf_counter(); c = _lib.Crf(ctx, rgb, 1, H, W, 3.0, 50.0, 5.0); ctx.sync(); t1 = time.perf_counter()
    c.close()
    t2 = time.perf_counter(); c = _lib.Crf(ctx, rgb, 1, H, W, 3.0, 50.0, 5.0); ctx.sync(); t3 = time.perf_counter()
    print(H, W, "first create %.2f ms, cached %.2f ms, on chip M=3: %s" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, c.gaussian_on_chip(3)))
    c.close()
