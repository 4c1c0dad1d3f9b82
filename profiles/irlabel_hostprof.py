"""cProfile of cam_to_ir_label.ir_label_batch (16 VOC-sized images, K = 2): where the host time goes."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, "wsss-analysis_amd")
from wsscam import _lib, synth
from wsscam.step import cam_to_ir_label as c2l
ctx = _lib.Context(0)
rng = np.random.default_rng(1)
B, H, W, K = 16, 375, 500, 2
imgs = np.stack([synth.synth_image(rng, H, W) for _ in range(B)])
maps = rng.random((B, K, H, W)).astype(np.float32)
keys = [np.array([3, 11])] * B
step = lambda: c2l.ir_label_batch(ctx, imgs, maps, keys, "voc12", 0.30, 0.05)
for _ in range(2): step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(3): step()
ctx.sync()
print("ms/image", (time.perf_counter() - t0) / 3 / B * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
ctx.sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
