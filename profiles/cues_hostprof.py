"""cProfile of cues.demo.gen_cues (VOC2012, VGG16 fg + bg models, 16 images of 321 x 321): where the host time goes."""
import cProfile, pstats, sys, time, tempfile
import numpy as np
sys.path.insert(0, "wsss-analysis_amd")
from wsscam import _lib, synth
from wsscam.cues import demo as cues_demo, utilities as cu
from wsscam.net import vgg16_cam
C = 20
def model(seed):
    sd = synth.plain_state_dict("vgg16", C, True, seed=seed)
    m = vgg16_cam.CAM(None, "voc12", "VGG16", C, None); m.load_state_dict(sd); m.cuda(0)
    return m
fg, bg = model(1), model(2)
alphas = {k: cu.get_grad_cam_weights(m, None, np.zeros((1, 321, 321, 3))) for k, m in (("fg", fg), ("bg", bg))}
thr = {"fg": np.full((1, C), 0.45), "bg": np.full((1, C), 0.45)}
rng = np.random.default_rng(3)
images = [synth.synth_image(rng, 321, 321) for _ in range(16)]
labels = (rng.random((16, C)) < 0.3).astype(np.float64)
out = tempfile.mkdtemp()
step = lambda: cues_demo.gen_cues("VOC2012", "VGG16", 0.2, 16, models={"fg": fg, "bg": bg}, alphas=alphas, thresholds=thr, images=images,
                                  labels=labels, out_dir=out, is_verbose=False)
step(); t0 = time.perf_counter(); step(); print("ms/image", (time.perf_counter() - t0) / 16 * 1e3)
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
