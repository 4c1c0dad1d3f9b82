"""segment -- mirror of the VOC2012 / DeepGlobe branch of 03c_hsn/demo.py:18-268 (HistoSegNet on natural images):
per batch, scores -> thresholds collapsed to 1/3 (demo.py:83) -> HSN Grad-CAM upsampled to the input
size -> foreground/background combination (VOC: bg channel 0.15 * expit(max_batch(X_bg) - X_bg), Q6) -> dense CRF
-> label maps.  `segment(dataset, model_type, batch_size)` alone loads models and the evaluation image list from settings.ini's
MODEL_ROOT / DATA_ROOT like the reference (wsscam.keras_store); callers that hold the objects pass them keyword-only:
  models {'fg': CAM wrapper[, 'bg': CAM wrapper]}, alphas {'fg': (F,C)[, 'bg']}, images: list of uint8 RGB.
Returns the list of (S, S) int64 label maps (what the reference feeds its evaluation / image writers)."""
import math
import os

import numpy as np
import scipy.special

from ..cues import utilities as cu
from ..cues.demo import read_batch_u8
from . import utilities as hu


def dcrf_config_for(dataset, model_type):
    """demo.py:157-165."""
    if dataset == "VOC2012" and model_type == "M7":
        return np.array([3 / 12 / 4, 3, 80 / 12 / 4, 13, 10, 10])
    return np.array([3 / 2, 3, 80 / 2, 13, 10, 10])


def segment(dataset, model_type, batch_size, set_name=None, should_saveimg=True, is_verbose=True, *, models=None, alphas=None,
            images=None, n_seg_classes=None, settings=None, reference_normalize_quirk=None):
    """`reference_normalize_quirk`: the reference's VOC2012 normalisation (03c_hsn/utilities.py:142-146) runs
    `x[:, :, 0] -= 104; x[:, :, 1] -= 117; x[:, :, 2] -= 123` on the 4-D uint8 BATCH -- that indexes image COLUMNS 0..2 of
    every row and channel, wraps around in uint8, and works in place, so the array later handed to the CRF is modified
    too; the result is divided by 255.  This arithmetic is the DEFAULT (True): the package is a drop-in, its VOC2012 label
    maps are the upstream ones.  False selects the per-channel (x - [104, 117, 123]) / 255 on an untouched image that the
    author evidently intended (what the 02_cues twin does, 02_cues/demo.py:163-164).  None (default) reads the optional key
    `hsn_voc_normalize = reference | intended` of settings.ini's [Data Folders] section when a settings file is in play,
    else True (deviations).

    Device resident between the batch upload and the label maps, like segment_adp: both models' Grad-CAM stacks are
    written straight into one [B][1 + C][S*S] stack (wsc_hsn_gradcam_post with a channel offset), the VOC background
    channel comes from wsc_hsn_voc_background (max over the whole batch, Q6), class mass flags from wsc_hsn_class_mass,
    unaries from wsc_hsn_gather_unary; the host sees scores, flags and labels."""
    from .. import _lib

    assert dataset in ["VOC2012", "DeepGlobe", "DeepGlobe_balanced"], "ADP: segment_adp"
    assert model_type in ["VGG16", "M7"]
    img_size = 321 if model_type == "VGG16" else 224
    from_settings = models is None or images is None
    if from_settings:
        # the reference's own call form segment(dataset, model_type, batch_size, ...): models, thresholds and the evaluation
        # image list come from settings.ini and the files under it (03c_hsn/demo.py:46-90), through keras_store
        from .. import keras_store as ks
        from ..cues.demo import _LazyImages

        st = ks.read_settings(settings)
        sess_id = dataset + "_" + model_type
        model_dir = os.path.join(st["MODEL_ROOT"], sess_id)
        if images is None:
            ds = ks.Dataset(data_type=dataset, size=img_size, batch_size=batch_size, database_dir=st["DATA_ROOT"], layout="hsn")
            images = _LazyImages(ds.set_gens[ds.sets[ds.is_evals.index(True)]])
        if models is None:
            models, alphas = {}, {}
            for m in (["fg", "bg"] if dataset == "VOC2012" else ["fg"]):  # demo.py:80-85: one directory, both modes
                models[m], alphas[m], _, _ = ks.load_model(model_dir, sess_id, model_type, dataset)
    voc = dataset == "VOC2012"
    if reference_normalize_quirk is None:
        # a settings file is consulted only when one is in play (passed, or the models / images were loaded through it above):
        # a call with everything in memory must not depend on a settings.ini relative to the working directory
        reference_normalize_quirk = True
        if settings is not None or from_settings:
            from .. import keras_store as ks

            reference_normalize_quirk = ks.read_option(settings, "hsn_voc_normalize", "reference") != "intended"
    if voc and is_verbose:
        print("\tVOC2012 normalisation: %s" % ("the reference's uint8 / column arithmetic (03c_hsn/utilities.py:142-146)"
                                              if reference_normalize_quirk else "per channel (x - [104, 117, 123]) / 255"))
    mean, std = ([104, 117, 123], [255, 255, 255]) if voc else ([0, 0, 0], [255, 255, 255])
    cfg = dcrf_config_for(dataset, model_type)
    out = []
    N = img_size * img_size
    n_batches = math.ceil(len(images) / batch_size)
    ctx = models["fg"].ctx
    thr_of = lambda a: np.full((1, np.asarray(a).shape[1]), 1.0 / 3.0)  # max(min(thr, 0), 1/3): demo.py:83, Q5
    for ib in range(n_batches):
        lo, hi = ib * batch_size, min((ib + 1) * batch_size, len(images))
        B = hi - lo
        chunk = images[lo:hi]
        # read_batch (03c_hsn/utilities.py:170-181): the batch is uint8 -- cv2.resize's 8-bit result -- and both the
        # normalised network input and the CRF image are made from it; resize on the device, normalisation too
        raw_u8 = read_batch_u8(chunk, (img_size, img_size), ctx=ctx)
        b_mean = mean  # per batch: the quirk path normalises the modified uint8 batch with a zero mean
        if voc and reference_normalize_quirk:
            raw_u8 = np.ascontiguousarray(raw_u8)
            raw_u8[:, :, 0] -= 104  # (B, H, 3): image column 0 of every row and channel, uint8 wrap-around, in place
            raw_u8[:, :, 1] -= 117
            raw_u8[:, :, 2] -= 123
            b_mean = [0, 0, 0]
        C = np.asarray(alphas["fg"]).shape[1]
        if voc:
            Cv = n_seg_classes or C + 1
            y_dev = ctx.alloc(B * Cv * N * 4, pooled=True)
            kw = dict(raw_u8=raw_u8, mean_std=(b_mean, std))
            hu.grad_cam_device(models["fg"], alphas["fg"], None, thr_of(alphas["fg"]), [img_size, img_size], out=(y_dev, Cv, 1),
                               ctx=ctx, **kw)
            Cb = np.asarray(alphas["bg"]).shape[1]
            hb = hu.grad_cam_device(models["bg"], alphas["bg"], None, thr_of(alphas["bg"]), [img_size, img_size], ctx=ctx, **kw)
            _lib.hsn_voc_background(ctx, hb[0], B, Cb, N, y_dev, Cv)
            valid = list(range(Cv))
        else:
            Cv = C
            kw = dict(raw_u8=raw_u8, mean_std=(b_mean, std))
            y_dev = hu.grad_cam_device(models["fg"], alphas["fg"], None, thr_of(alphas["fg"]), [img_size, img_size], ctx=ctx,
                                       **kw)[0]
            valid = list(range(C - 1))  # Y = H_fg[:, :-1]: the 'unknown' class is dropped (demo.py:153)
        mass_dev = ctx.alloc(B * Cv * 4, pooled=True)
        _lib.hsn_class_mass(ctx, y_dev, B * Cv, N, mass_dev)
        mass = ctx.to_host(mass_dev, (B, Cv), np.uint32)
        mass[:, [c for c in range(Cv) if c not in valid]] = 0
        out.extend(list(hu.dcrf_process_device(ctx, y_dev, mass, raw_u8, Cv, img_size, img_size, cfg)))
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


def adipose_source_channels(ac):
    """Channels of the all-class Grad-CAM stack that the reference's `Y_gradcam['morph'][:, adipose_inds]` reads
    (demo.py:368-369; 02_cues/demo.py:307-308): positions of A.W / A.B / A.M in classes['morph'], used as channel numbers of the
    valid morph stack, traced back through morph2valid / all2morph."""
    morph_src = dict(zip(ac.classinds["morph2valid"], ac.classinds["all2morph"]))
    return [morph_src[i] for i, x in enumerate(ac.classes["morph"]) if x in ["A.W", "A.B", "A.M"]]


def lane_contexts(model, n_lanes):
    """Contexts (= HIP streams with their own workspace and buffer pool) of `n_lanes` batches in flight on the model's device:
    the model's own context, then extras created once and kept with the model."""
    from .. import _lib

    extra = model.__dict__.setdefault("_lane_ctxs", [])
    while len(extra) < n_lanes - 1:
        extra.append(_lib.Context(model.ctx.device))
    return [model.ctx] + extra[:n_lanes - 1]


def run_batches_on_lanes(n_batches, ctxs, one_batch):
    """one_batch(batch index, ctx) for every batch; lane k (a thread with context ctxs[k]) takes batches k, k + L, ... one after
    the other, so a batch's host decisions (scores -> gate, class mass -> CRF set-up: a handful of small read-backs, each a
    stream synchronisation) are hidden behind the other lane's kernels -- the reference's loop (03c_hsn/demo.py:318-380) is
    serial and the device sat idle 15-20 % of a batch.  Results in batch order."""
    if n_batches <= 1 or len(ctxs) <= 1:
        return [one_batch(b, ctxs[0]) for b in range(n_batches)]
    from concurrent.futures import ThreadPoolExecutor

    out = [None] * n_batches

    def lane(k):
        for b in range(k, n_batches, len(ctxs)):
            out[b] = one_batch(b, ctxs[k])

    with ThreadPoolExecutor(len(ctxs)) as ex:
        for f in [ex.submit(lane, k) for k in range(min(len(ctxs), n_batches))]:
            f.result()
    return out


def segment_adp(model, alpha, thresholds, images, dcrf_configs, size, batch_size, is_verbose=False, all_classes=None, stats=None,
                n_lanes=3, chain_stacks=False):
    """demo.py:271-380 for one ADP model: per batch scores >= thresholds -> HSN Grad-CAM at (size, size) -> per
    HTT type {morph, func}: scatter into the valid-class stack, modify_by_htt (background / other channels),
    get_cs_gradcam, dense CRF with that type's configuration.  `images` are uint8 RGB (any size; resized like
    ADPCues.read_batch), `dcrf_configs` {'morph': 6-vector, 'func': 6-vector} (the reference reads
    `<htt>_optimal_pcc.npy`).  Returns {'morph': [label maps], 'func': [label maps]} at (size, size).

    Device resident between the batch upload and the label maps: wsc_net_forward_gradcam -> wsc_hsn_gradcam_post ->
    wsc_hsn_background -> wsc_hsn_cs_gradcam -> wsc_hsn_gather_unary -> wsc_crf_*; the host sees the (B, C) scores, the
    (B, Cv) class-mass flags and the final labels.  `stats` (optional dict): per HTT type the list of every image's number of
    classes with mass -- the M its dense CRF ran with (dcrf_process keeps the classes whose maps are not all zero, :425).
    `n_lanes` batches are in flight at once, each on its own stream (run_batches_on_lanes); a batch's results do not depend
    on the lane it ran on.  `chain_stacks`: the lanes' VGG16 passes take turns on the device (_lib.StackChain) -- measured neutral
    here (1067-1140 against 1053-1181 images/s, profiles/r06_chain_ab.txt: a batch is mostly CRF), so off by default."""
    from .. import _lib

    ac = ADPClasses(all_classes)
    N = size * size
    C_all = len(ac.classes["all"])
    # demo.py:368-369: `adipose_inds` are positions of A.W / A.B / A.M in classes['morph'] (18, 19, 20) but index
    # Y_gradcam['morph'], the VALID stack whose channel 0 is 'Background' -- so the reference's "adipose" maps are the valid
    # stack's channels 18..20 = S.R, A.W, A.B (off by one; the 02_cues twin has the same line, demo.py:307-308).  Reproduced:
    # those channels of the valid stack, traced back to their source channels of the all-class stack (a quirk found in round 5).
    adipose_all = adipose_source_channels(ac)
    bounds = [(lo, min(lo + batch_size, len(images))) for lo in range(0, len(images), batch_size)]
    model.gradcam_net(np.asarray(alpha))  # (built once, before the lanes' threads ask for it)
    # several batches in flight: their VGG16 stacks take turns on the device, everything else of a batch overlaps (_lib.StackChain)
    chain = model.__dict__.setdefault("_stack_chain", _lib.StackChain()) if int(n_lanes) > 1 and chain_stacks else None

    def one_batch(bi, ctx):
        lo, hi = bounds[bi]
        B = hi - lo
        chunk = images[lo:hi]
        res = {"morph": None, "func": None, "stats": {}}
        raw = read_batch_u8(chunk, (size, size), ctx=ctx)  # ADPCues.read_batch: cv2.resize's uint8 batch (adp_cues.py:122-128)
        # (raw - 193.09203) / 56.450138 (adp_cues.py:130) and the NHWC -> NCHW layout on the device
        H_dev, scores, is_pass, _, raw_dev = hu.grad_cam_device(model, alpha, None, thresholds, [size, size], raw_u8=raw,
                                                                mean_std=(193.09203, 56.450138), ctx=ctx, chain=chain)
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
            res["stats"][htt] = [int(v) for v in (mass > 0).sum(1)]
            res[htt] = list(hu.dcrf_process_device(ctx, cs_dev, mass, raw, Cv, size, size, dcrf_configs[htt]))
        if is_verbose:
            print("\tBatch %d-%d" % (lo, hi))
        return res

    out = {"morph": [], "func": []}
    for res in run_batches_on_lanes(len(bounds), lane_contexts(model, max(1, int(n_lanes))), one_batch):
        for htt in ("morph", "func"):
            out[htt].extend(res[htt])
            if stats is not None:
                stats.setdefault(htt, []).extend(res["stats"][htt])
    return out


# ---- evaluation tail of the HistoSegNet drivers (03c_hsn/demo.py:386-408, 424-428; VOC / DeepGlobe: :176-197) -----------
def gt_index_from_colours(gt_rgb, colours):
    """Colour-coded ground truth (H, W, 3) -> uint8 class index; a pixel of no listed colour gets len(colours) -- the
    reference tests `gt == colour k` per class (demo.py:392-397), so such a pixel is in no ground-truth mask but still
    counts in the unions through the prediction masks."""
    gt_rgb = np.asarray(gt_rgb, dtype=np.uint8)
    idx = np.full(gt_rgb.shape[:2], len(colours), dtype=np.uint8)
    for k in range(len(colours) - 1, -1, -1):  # (distinct colours; the first listed class wins if two ever coincide)
        idx[np.all(gt_rgb == np.asarray(colours[k], dtype=np.uint8)[None, None, :], axis=2)] = k
    return idx


class LabelEvaluator:
    """Carries one HTT type's confusion matrix on the device over a whole run and turns it into the reference's numbers.

    update(label_maps, gt_rgb): demo.py:386-408 for a batch -- cv2.resize(pred, (1088, 1088), INTER_NEAREST) of every
    (S, S) label map and the per-class counts, as ONE pass per batch on the device (wsc_label_confusion_nn).
    metrics(): intersects / unions / gt_count / confusion_matrix / IoU per class / mIoU exactly as :399-405 and :424-428
    define them (IoU = intersect / (union + 1e-7), mIoU = their mean over ALL valid classes)."""

    def __init__(self, ctx, colours, out_size=(1088, 1088)):
        self.ctx, self.colours, self.out_size = ctx, [tuple(int(v) for v in c) for c in colours], tuple(out_size)
        self.n = len(self.colours)
        self.conf_dev = ctx.alloc((self.n + 1) * (self.n + 1) * 8)
        from .. import _lib

        _lib.check(ctx._lib.wsc_memset(ctx.h, self.conf_dev.ptr, 0, (self.n + 1) * (self.n + 1) * 8))

    def update(self, label_maps, gt_rgb, want_pred=False):
        from .. import _lib

        B = len(label_maps)
        lab = np.ascontiguousarray(np.stack([np.asarray(m) for m in label_maps]).astype(np.int32))
        h, w = lab.shape[1:]
        gt = np.ascontiguousarray(np.stack([gt_index_from_colours(g, self.colours) for g in gt_rgb]))
        assert gt.shape == (B,) + self.out_size, (gt.shape, self.out_size)
        lab_dev, gt_dev = self.ctx.to_device(lab, pooled=True), self.ctx.to_device(gt, pooled=True)
        pred_dev = self.ctx.alloc(B * self.out_size[0] * self.out_size[1], pooled=True) if want_pred else None
        _lib.label_confusion_nn(self.ctx, lab_dev, [(h, w)] * B, [self.out_size] * B, [b * h * w for b in range(B)], gt_dev,
                                self.n + 1, self.conf_dev, pred_dev=pred_dev, ignore_label=255)
        if want_pred:
            return self.ctx.to_host(pred_dev, (B,) + self.out_size, np.uint8)

    def metrics(self):
        conf = self.ctx.to_host(self.conf_dev, (self.n + 1, self.n + 1), np.int64)
        n = self.n
        inter = np.diag(conf)[:n].astype(np.float64)
        gt_count = conf[:n, :].sum(1).astype(np.float64)
        pred_count = conf[:, :n].sum(0).astype(np.float64)  # over every pixel, whatever its ground-truth colour
        union = gt_count + pred_count - inter
        iou = inter / (union + 1e-7)
        return {"confusion_matrix": conf[:n, :n].astype(np.float64), "intersects": inter, "unions": union, "gt_count": gt_count,
                "IoU": iou, "mIoU": float(np.mean(iou))}
