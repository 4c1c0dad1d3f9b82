"""cProfile of one HSN step (bench.py --workload hsn): where the host time between the kernels goes."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, "wsss-analysis_amd")
from wsscam import _lib, synth
from wsscam.hsn import demo as hsn_demo
from wsscam.net import vgg16_cam
from wsscam.net.common import grad_cam_alpha
C, S_ = 31, 321
sd = synth.plain_state_dict("vgg16", C, batchnorm=False, seed=0)
model = vgg16_cam.CAM(None, "adp_morph", "ADP_VGG16", C, None, precision=_lib.PREC_F16)
model.load_state_dict(sd); model.cuda(0)
alpha = grad_cam_alpha(sd["vgg16.classifier.0.weight"], S_ // 8, S_ // 8, "avg")
rng = np.random.default_rng(4242)
images = [synth.adp_image(rng, S_, S_) for _ in range(16)]
thr = np.full((1, C), 0.5)
cfgs = {"morph": np.array([3 / 2, 3, 80 / 2, 13, 10, 10]), "func": np.array([3 / 2, 3, 80 / 2, 13, 10, 10])}
step = lambda: hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S_, 16)
for _ in range(3): step()
model.ctx.sync()
t0 = time.perf_counter()
for _ in range(5): step()
model.ctx.sync()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
model.ctx.sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
