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


# ---- ADP (03c_hsn/demo.py:271-407 with the class bookkeeping of 03c_hsn/adp_cues.py:20-58) --------------------------
ADP_MORPH = ["E.M.S", "E.M.U", "E.M.O", "E.T.S", "E.T.U", "E.T.O", "E.P", "C.D.I", "C.D.R", "C.L", "H.E", "H.K", "H.Y",
             "S.M.C", "S.M.S", "S.E", "S.C.H", "S.R", "A.W", "A.B", "A.M", "M.M", "M.K", "N.P", "N.R.B", "N.R.A", "N.G.M",
             "N.G.W"]
ADP_FUNC = ["G.O", "G.N", "T"]


class ADPClasses:
    """classes / classinds of ADPCues.__init__ for the 31-class (non-X1.7) models."""

    def __init__(self, all_classes=None):
        self.classes = {"all": list(all_classes) if all_classes is not None else ADP_MORPH + ADP_FUNC, "morph": ADP_MORPH,
                        "func": ADP_FUNC, "valid_morph": ["Background"] + ADP_MORPH,
                        "valid_func": ["Background", "Other"] + ADP_FUNC}
        c = self.classes
        self.classinds = {
            "morph2valid": [i for i, x in enumerate(c["valid_morph"]) if x in c["morph"]],
            "func2valid": [i for i, x in enumerate(c["valid_func"]) if x in c["func"]],
            "all2morph": [i for i, x in enumerate(c["all"]) if x in c["valid_morph"]],
            "all2func": [i for i, x in enumerate(c["all"]) if x in c["valid_func"]],
        }


def segment_adp(model, alpha, thresholds, images, dcrf_configs, size, batch_size, is_verbose=False, all_classes=None):
    """demo.py:271-380 for one ADP model: per batch scores >= thresholds -> HSN Grad-CAM at (size, size) -> per
    HTT type {morph, func}: scatter into the valid-class stack, modify_by_htt (background / other channels),
    get_cs_gradcam, dense CRF with that type's configuration.  `images` are uint8 RGB (any size; resized like
    ADPCues.read_batch), `dcrf_configs` {'morph': 6-vector, 'func': 6-vector} (the reference reads
    `<htt>_optimal_pcc.npy`).  Returns {'morph': [label maps], 'func': [label maps]} at (size, size).

    Device resident between the batch upload and the label maps: wsc_net_forward_gradcam -> wsc_hsn_gradcam_post ->
    wsc_hsn_background -> wsc_hsn_cs_gradcam -> wsc_hsn_gather_unary -> wsc_crf_*; the host sees the (B, C) scores, the
    (B, Cv) class-mass flags and the final labels."""
    from .. import _lib

    ac = ADPClasses(all_classes)
    out = {"morph": [], "func": []}
    N = size * size
    C_all = len(ac.classes["all"])
    adipose_all = [i for i, x in enumerate(ac.classes["all"]) if x in ["A.W", "A.B", "A.M"]]
    for lo in range(0, len(images), batch_size):
        hi = min(lo + batch_size, len(images))
        B = hi - lo
        chunk = images[lo:hi]
        if all(np.asarray(im).shape == (size, size, 3) and np.asarray(im).dtype == np.uint8 for im in chunk):
            raw = np.ascontiguousarray(np.stack(chunk))       # already at the network size: nothing to resize
        else:
            _, raw = read_batch(chunk, (size, size), [0, 0, 0], [1, 1, 1])
            raw = np.clip(np.rint(raw), 0, 255).astype(np.uint8)  # ADPCues.read_batch keeps the resized batch as uint8
        # (raw - 193.09203) / 56.450138 (adp_cues.py:130) and the NHWC -> NCHW layout on the device
        H_dev, scores, is_pass, ctx, raw_dev = hu.grad_cam_device(model, alpha, None, thresholds, [size, size], raw_u8=raw,
                                                                  mean_std=(193.09203, 56.450138))
        bg_dev = ctx.alloc(B * N * 8, pooled=True)
        _lib.hsn_background(ctx, raw_dev, B, size, size, bg_dev)
        for htt in ("morph", "func"):
            valid = ac.classes["valid_" + htt]
            Cv = len(valid)
            src_of = [-1] * Cv
            for v, a_ in zip(ac.classinds[htt + "2valid"], ac.classinds["all2" + htt]):
                src_of[v] = a_
            bg_ind, other_ind, ex_inds = hu._htt_tables(valid, htt == "func")
            cs_dev, mass_dev = ctx.alloc(B * Cv * N * 4, pooled=True), ctx.alloc(B * Cv * 4, pooled=True)
            _lib.hsn_cs_gradcam(ctx, H_dev, B, C_all, N, bg_dev, src_of, bg_ind, other_ind, ex_inds,
                                adipose_all if htt == "func" else None, cs_dev, None, mass_dev)
            mass = ctx.to_host(mass_dev, (B, Cv), np.uint32)
            out[htt].extend(list(hu.dcrf_process_device(ctx, cs_dev, mass, raw, Cv, size, size, dcrf_configs[htt])))
        if is_verbose:
            print("\tBatch %d-%d" % (lo, hi))
    return out
