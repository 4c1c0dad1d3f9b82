"""segment -- mirror of the VOC2012 / DeepGlobe branch of 03c_hsn/demo.py:18-268 (HistoSegNet on natural images):
per batch, scores -> thresholds collapsed to 1/3 (demo.py:83) -> HSN Grad-CAM upsampled to the input
size -> foreground/background combination (VOC: bg channel 0.15 * expit(max_batch(X_bg) - X_bg), Q6) -> dense CRF
-> label maps.  The Keras model / dataset objects the reference builds from MODEL_ROOT / DATA_ROOT are passed in:
  models {'fg': CAM wrapper[, 'bg': CAM wrapper]}, alphas {'fg': (F,C)[, 'bg']}, images: list of uint8 RGB.
Returns the list of (S, S) int64 label maps (what the reference feeds its evaluation / image writers)."""
import math

import numpy as np
import scipy.special

from ..cues import utilities as cu
from ..cues.demo import read_batch
from . import utilities as hu


def dcrf_config_for(dataset, model_type):
    """demo.py:157-165."""
    if dataset == "VOC2012" and model_type == "M7":
        return np.array([3 / 12 / 4, 3, 80 / 12 / 4, 13, 10, 10])
    return np.array([3 / 2, 3, 80 / 2, 13, 10, 10])


def segment(dataset, model_type, batch_size, set_name=None, should_saveimg=True, is_verbose=True, *, models, alphas,
            images, n_seg_classes=None):
    assert dataset in ["VOC2012", "DeepGlobe", "DeepGlobe_balanced"], "ADP: segment_adp"
    assert model_type in ["VGG16", "M7"]
    img_size = 321 if model_type == "VGG16" else 224
    fgbg_modes = ["fg", "bg"] if dataset == "VOC2012" else ["fg"]
    mean, std = ([104, 117, 123], [255, 255, 255]) if dataset == "VOC2012" else ([0, 0, 0], [255, 255, 255])
    cfg = dcrf_config_for(dataset, model_type)
    out = []
    n_batches = math.ceil(len(images) / batch_size)
    for ib in range(n_batches):
        lo, hi = ib * batch_size, min((ib + 1) * batch_size, len(images))
        norm, raw = read_batch(images[lo:hi], (img_size, img_size), mean, std)
        H = {}
        for m in fgbg_modes:
            _, scores = cu.conv_and_cams(models[m], np.asarray(alphas[m]), norm, relu=False, want_scores=True)
            thr = np.full((1, scores.shape[1]), 1.0 / 3.0)  # max(min(thr, 0), 1/3)
            is_pass = np.greater_equal(scores, thr)
            g = hu.grad_cam(models[m], alphas[m], norm, is_pass, "final", scores, orig_sz=[img_size, img_size],
                            should_upsample=True)
            H[m] = np.transpose(g, (0, 3, 1, 2))
        if dataset == "VOC2012":
            C = H["fg"].shape[1]
            Y = np.zeros((hi - lo, (n_seg_classes or C + 1), img_size, img_size))
            X_bg = np.sum(H["bg"], axis=1)
            Y[:, 0] = 0.15 * scipy.special.expit(np.max(X_bg) - X_bg)  # max over the whole batch (Q6)
            Y[:, 1:] = H["fg"]
        else:
            Y = H["fg"][:, :-1, :, :]
        out.extend(list(hu.dcrf_process(Y, raw.astype(np.uint8), cfg, ctx=models["fg"].ctx)))
        if is_verbose:
            print("\tBatch #%d of %d" % (ib + 1, n_batches))
    return out
