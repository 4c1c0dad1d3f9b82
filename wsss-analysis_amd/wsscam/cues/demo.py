"""gen_cues -- mirror of 02_cues/demo.py:26-222 (VOC2012 / DeepGlobe seed generation; the ADP redirect uses
cues.utilities.update_cues_adp on hsn.utilities.modify_by_htt's output).

Same signature: `gen_cues(dataset, model_type, thresh, batch_size, set_name, run_train, is_verbose)` alone reads what the
reference reads -- settings.ini, the session's `.h5` weights + `.mat` thresholds under MODEL_ROOT (the fixed device
architectures stand in for the `.json`), the split's CSV and images under DATA_ROOT -- through wsscam.keras_store and
writes the pickle under the cues root.  Callers that hold the loaded objects pass them keyword-only instead:
  models     {'fg': CAM wrapper, 'bg': CAM wrapper}  (wsscam.net.vgg16_cam.CAM / m7_cam.CAM with weights loaded)
  alphas     {'fg': (F,C), 'bg': (F,C)} Grad-CAM weights (cues.utilities.get_grad_cam_weights or precomputed)
  thresholds {'fg': (1,C), 'bg': (1,C)}  optimalScoreThresh of the .mat files
  images     uint8 RGB images (list of (H,W,3)), `labels` (n, C) image-level labels (gen_curr.data)
  out_dir    where `localization_cues.pickle` / `localization_cues_val.pickle` is written
Per batch: scores + Grad-CAMs of the fg (and, for VOC2012, bg) model in ONE device pass each (the reference runs
model.predict and K.function separately), resize to the 41 x 41 seed size on the device, threshold / overlap
resolution on the host, pickle in the layout 03a_sec-dsrg/model.py:238-246 reads."""
import math
import os
import pickle

import numpy as np

from . import utilities as cu

SEED_SIZE = 41


class _LazyImages:
    """The images of a keras_store.SetList, decoded per batch slice (len + slicing is all the batch loop needs)."""

    def __init__(self, set_list):
        self.set_list = set_list

    def __len__(self):
        return len(self.set_list.filenames)

    def __getitem__(self, sl):
        if isinstance(sl, slice):
            return self.set_list.images(sl.start or 0, sl.stop)
        return self.set_list.images(sl, sl + 1)[0]


def read_batch_u8(images, size, ctx=None):
    """The uint8 batch of read_batch: `img_batch = np.empty(..., dtype='uint8'); img_batch[i] = cv2.resize(tmp, size)`
    (02_cues/utilities.py:172-176, 02_cues/adp_cues.py / 03c_hsn/adp_cues.py:122-128, 03c_hsn/utilities.py:170-181) -- OpenCV's
    8-bit INTER_LINEAR result, which the network input AND the CRF image are made of.  With `ctx` the resize runs on the
    device (wsc_resize_u8: the decoded images travel once, packed), else in numpy (voc12.dataloader.resize_bilinear_u8);
    the two are bit-identical.  Images already at `size` are copied (the reference's ADP twin does that, :127-128; its
    VOC / DeepGlobe twin leaves them uninitialised)."""
    from ..voc12.dataloader import resize_bilinear_u8

    ims = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
    B, (oh, ow) = len(ims), (int(size[0]), int(size[1]))
    if ctx is None or all(im.shape[:2] == (oh, ow) for im in ims):
        return np.stack([resize_bilinear_u8(im, (oh, ow)) for im in ims]) if B else np.empty((0, oh, ow, 3), np.uint8)
    from .. import _lib

    offs = np.concatenate(([0], np.cumsum([im.size for im in ims]))).astype(np.int64)
    packed = np.concatenate([im.reshape(-1) for im in ims])
    src_dev, out_dev = ctx.to_device(packed, pooled=True), ctx.alloc(B * oh * ow * 3, pooled=True)
    _lib.resize_u8(ctx, src_dev, [im.shape[:2] for im in ims], offs[:-1], (oh, ow), out_dev)
    return ctx.to_host(out_dev, (B, oh, ow, 3), np.uint8)


def read_batch(images, size, img_mean, img_std, ctx=None):
    """02_cues/utilities.py:146-181 on in-memory images: (normalised float64 (B,S,S,3), uint8 batch (B,S,S,3)); the
    normalisation starts from the uint8 batch like the reference's (:177-180)."""
    raw = read_batch_u8(images, size, ctx=ctx)
    norm = (raw - np.asarray(img_mean, np.float64)) / np.asarray(img_std, np.float64)
    return norm, raw


def gen_cues(dataset, model_type, thresh, batch_size, set_name=None, run_train=True, is_verbose=True, *, models=None,
             alphas=None, thresholds=None, images=None, labels=None, out_dir=None, class_names=None, settings=None):
    assert dataset in ["VOC2012", "DeepGlobe", "DeepGlobe_balanced"], "ADP: use gen_cues_adp"
    assert model_type in ["X1.7", "M7", "VGG16"]
    assert batch_size > 0 and set_name in [None, "tuning", "segtest"]
    img_size = 321 if model_type in ["VGG16", "VGG16bg"] else 224
    fgbg_modes = ["fg", "bg"] if dataset == "VOC2012" else ["fg"]
    if models is None or images is None or out_dir is None:
        # the reference's own call form gen_cues(dataset, model_type, thresh, batch_size, ...): everything else comes from
        # settings.ini and the files under it (02_cues/demo.py:16-24, 60-150), through keras_store
        from .. import keras_store as ks

        st = ks.read_settings(settings)
        sess_id = dataset + "_" + model_type if set_name is None else dataset + "_" + set_name + "_" + model_type
        if thresh != 0.2:
            sess_id += "_" + str(thresh)
        model_dir = os.path.join(st["MODEL_ROOT"], dataset + "_" + model_type)
        if out_dir is None:
            out_dir = os.path.join(st["CUES_ROOT"], sess_id) if run_train else os.path.join("./eval", sess_id)
        if images is None:
            # 02_cues/dataset.py:12 ignores settings.ini: <parent of cwd>/database.  An explicitly passed settings file is a
            # caller with its own tree, whose data_dir is honoured.
            ds = ks.Dataset(data_type=dataset, size=img_size, batch_size=batch_size,
                            database_dir=st["DATA_ROOT"] if settings is not None else None)
            gen_curr = ds.set_gens[ds.sets[ds.is_evals.index(not run_train)]]
            images, labels = _LazyImages(gen_curr), gen_curr.data
        if models is None:
            models, alphas, thresholds = {}, {}, {}
            for m, (mdir, sid) in ks.fgbg_sessions(model_dir, sess_id, fgbg_modes).items():
                models[m], alphas[m], _, thresholds[m] = ks.load_model(mdir, sid, model_type, dataset)
    labels = np.asarray(labels)
    n_cls = labels.shape[1]
    if dataset == "VOC2012":
        mean, std, ignore_ind = [104, 117, 123], [255, 255, 255], None
    else:
        mean, std, ignore_ind = [0, 0, 0], [255, 255, 255], 6
    keep_inds = np.arange(n_cls)
    thr = {m: np.asarray(thresholds[m]).reshape(1, -1) for m in fgbg_modes}
    if ignore_ind is not None:
        keep_inds = np.delete(keep_inds, ignore_ind)
        thr = {m: t[:, keep_inds] for m, t in thr.items()}
    cues = {}
    n_batches = math.ceil(len(images) / batch_size)
    for ib in range(n_batches):
        lo, hi = ib * batch_size, min((ib + 1) * batch_size, len(images))
        if is_verbose:
            print("\tBatch #%d of %d" % (ib + 1, n_batches))
        norm, _ = read_batch(images[lo:hi], (img_size, img_size), mean, std, ctx=models["fg"].ctx)
        H, is_pass = {}, {}
        for m in fgbg_modes:
            cams, scores = cu.conv_and_cams(models[m], np.asarray(alphas[m]), norm, relu=True, want_scores=True)
            scores = scores[:, keep_inds]
            is_pass[m] = np.greater_equal(scores, thr[m]) * labels[lo:hi][:, keep_inds]
            cams = cams[:, :, :, keep_inds].astype(np.float64) * is_pass[m][:, None, None, :]  # grad_cam :135-144
            H[m] = cu.resize_stack(np.transpose(cams, (0, 3, 1, 2)), (SEED_SIZE, SEED_SIZE), ctx=models[m].ctx)
        idx = list(range(lo, hi))
        if dataset == "VOC2012":
            class_inds = [np.where(is_pass["fg"][i])[0] + 1 for i in range(hi - lo)]
            cues = cu.get_fgbg_cues(cues, H["fg"], H["bg"], class_inds, idx, thresh)
        else:
            class_inds = [np.where(is_pass["fg"][i])[0] for i in range(hi - lo)]
            cues = cu.get_fg_cues(cues, H["fg"], class_inds, idx, thresh)
    os.makedirs(out_dir, exist_ok=True)
    name = "localization_cues.pickle" if run_train else "localization_cues_val.pickle"
    with open(os.path.join(out_dir, name), "wb") as f:
        pickle.dump(cues, f)
    return cues


def gen_cues_adp(model_type, thresh, batch_size, size, cues_dir, set_name, is_verbose, *, model, alpha, thresholds,
                 images, all_classes=None):
    """02_cues/demo.py:224-310 (ADP seed generation): one model, scores >= thresholds, ReLU Grad-CAM, 41 x 41
    seeds, per HTT type {morph, func} the valid-class stack with the background / other channels of
    modify_by_htt and update_cues's per-image thresholds; writes `<cues_dir>/ADP-<htt>_.../localization_cues.pickle`
    like the reference when `cues_dir` is a dict {'morph': dir, 'func': dir}, else returns the two cue dicts."""
    from ..hsn import utilities as hu
    from ..hsn.demo import ADPClasses

    ac = ADPClasses(all_classes)
    arr = {k: np.array(ac.classinds[k]) for k in ("morph2valid", "func2valid")}
    morph_all = [i for i, x in enumerate(ac.classes["all"]) if x in ac.classes["morph"]]
    func_all = [i for i, x in enumerate(ac.classes["all"]) if x in ac.classes["func"]]
    cues = {"morph": {}, "func": {}}
    thr = np.asarray(thresholds).reshape(1, -1)
    for lo in range(0, len(images), batch_size):
        hi = min(lo + batch_size, len(images))
        raw = read_batch_u8(images[lo:hi], (size, size), ctx=model.ctx)  # ADPCues.read_batch keeps the batch as uint8
        norm = (raw - 193.09203) / 56.450138
        cams, scores = cu.conv_and_cams(model, np.asarray(alpha), norm, relu=True, want_scores=True)
        is_pass = np.greater_equal(scores, thr)
        H = cams.astype(np.float64) * is_pass[:, None, None, :]          # ADPCues.grad_cam, adp_cues.py:191-223
        H = cu.resize_stack(np.transpose(H, (0, 3, 1, 2)), (SEED_SIZE, SEED_SIZE), ctx=model.ctx)
        ip = {"morph": is_pass[:, morph_all], "func": is_pass[:, func_all]}
        seeds = {}
        for htt in ("morph", "func"):
            valid = ac.classes["valid_" + htt]
            seeds[htt] = np.zeros((hi - lo, len(valid), SEED_SIZE, SEED_SIZE))
            seeds[htt][:, ac.classinds[htt + "2valid"]] = H[:, ac.classinds["all2" + htt]]
            class_inds = [arr[htt + "2valid"][ip[htt][i]] for i in range(hi - lo)]
            if htt == "morph":
                seeds[htt] = hu.modify_by_htt(seeds[htt], raw, valid)
            else:
                class_inds = [np.append(1, x) for x in class_inds]
                adipose = [i for i, x in enumerate(ac.classes["morph"]) if x in ["A.W", "A.B", "A.M"]]
                seeds[htt] = hu.modify_by_htt(seeds[htt], raw, valid, gradcam_adipose=seeds["morph"][:, adipose])
            cu.update_cues_adp(cues[htt], seeds[htt], class_inds, list(range(lo, hi)), thresh)
        if is_verbose:
            print("\tBatch %d-%d" % (lo, hi))
    if isinstance(cues_dir, dict):
        for htt in ("morph", "func"):
            os.makedirs(cues_dir[htt], exist_ok=True)
            with open(os.path.join(cues_dir[htt], "localization_cues.pickle"), "wb") as f:
                pickle.dump(cues[htt], f)
    return cues


# ---- evaluation of the cues (02_cues/demo.py:323-484) -------------------------------------------------------------------------
VOC_SEG_CLASS_NAMES = ["__background__", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
                       "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def cue_label_map(cues_i, n_classes, seed_size=SEED_SIZE, empty_label=0):
    """demo.py:423-427 / :446-451: `cues_pred[cues_i[1], cues_i[2], cues_i[0]] = 1; argmax(cues_pred, -1)` -- the class of every
    seed pixel, `empty_label` where no class claims it (VOC2012: arg-max of an all-zero vector = 0, the background; DeepGlobe:
    `ignore_ind`, a class outside the scored ones)."""
    cues_i = np.asarray(cues_i)
    pred = np.zeros((seed_size, seed_size, n_classes))
    if cues_i.size:
        pred[cues_i[1], cues_i[2], cues_i[0]] = 1.0
    lab = np.argmax(pred, axis=-1).astype(np.int32)
    lab[np.sum(pred, axis=-1) == 0] = empty_label
    return lab


def cue_confusion(ctx, cues, gt_index_maps, n, empty_label, batch_size, is_verbose=False, first_index=0):
    """Confusion counts [gt][pred] of (n + 1) x (n + 1) classes (index n: 'no class') between the cue label maps -- arg-max of
    the one-hot seeds, cv2 nearest-neighbour resized to each ground truth's own size -- and the ground-truth index maps, on the
    device (wsc_label_confusion_nn).  Every quantity of 02_cues/demo.py:428-433 / :586-597 is a sum over this matrix."""
    from .. import _lib

    conf_dev = ctx.alloc((n + 1) * (n + 1) * 8)
    _lib.check(ctx._lib.wsc_memset(ctx.h, conf_dev.ptr, 0, (n + 1) * (n + 1) * 8))
    n_img = len(gt_index_maps)
    for lo in range(0, n_img, batch_size):
        hi = min(lo + batch_size, n_img)
        labs = [cue_label_map(cues["%d_cues" % (first_index + i)], n, empty_label=empty_label) for i in range(lo, hi)]
        gidx = [np.ascontiguousarray(gt_index_maps[i], dtype=np.uint8) for i in range(lo, hi)]
        lab_dev = ctx.to_device(np.ascontiguousarray(np.stack(labs)), pooled=True)
        gt_dev = ctx.to_device(np.concatenate([g.reshape(-1) for g in gidx]), pooled=True)
        _lib.label_confusion_nn(ctx, lab_dev, [(SEED_SIZE, SEED_SIZE)] * (hi - lo), [g.shape for g in gidx],
                                [b * SEED_SIZE * SEED_SIZE for b in range(hi - lo)], gt_dev, n + 1, conf_dev, ignore_label=-1)
        if is_verbose:
            print("\tImages %d-%d of %d" % (lo + 1, hi, n_img))
    return ctx.to_host(conf_dev, (n + 1, n + 1), np.int64)


def eval_cues(dataset, model_type, thresh, batch_size, set_name=None, run_train=False, should_saveimg=False, is_verbose=True, *,
              cues=None, gts=None, colours=None, class_names=None, out_dir=None, ctx=None, settings=None):
    """02_cues/demo.py:323-484 for VOC2012 / DeepGlobe: every image's cues -> 41 x 41 class map -> cv2 nearest-neighbour
    resize to the ground truth's size -> per-class intersections and unions over the whole set -> IoU = I / (U + 1e-7), mIoU
    = their mean; written as `metrics_<sess_id>_<set>.csv` (and `.xlsx` when openpyxl is installed, the reference's format).
    The resize and the counting run on the device (wsc_label_confusion_nn, the kernel of the HistoSegNet evaluation tail).
      cues     the dict gen_cues(..., run_train=False) returns / pickles (`'%d_cues'` -> int64 (3, n) rows class, row, col)
      gts      per image: VOC2012 the class-index map (H, W) uint8 (what `cv2.imread(png)[:, :, 0]` of the reference reads);
               DeepGlobe the colour-coded (H, W, 3) RGB map, matched against `colours`
    Both default to what the reference reads from disk: ./eval/<sess_id>/localization_cues_val.pickle and the PNGs under
    SegmentationClassAug next to the evaluation images (wsscam.keras_store).  Debug images (`should_saveimg`) are not
    rendered.  Returns {'intersects', 'unions', 'IoU', 'mIoU', 'classes'}."""
    from .. import _lib
    from ..misc.imutils import default_context

    assert dataset in ["VOC2012", "DeepGlobe", "DeepGlobe_balanced"], "ADP: eval_cues_adp is not mirrored"
    assert batch_size > 0 and set_name in [None, "tuning", "segtest"]
    sess_id = dataset + "_" + model_type if set_name is None else dataset + "_" + set_name + "_" + model_type
    if thresh != 0.2:
        sess_id += "_" + str(thresh)
    eval_dir = out_dir or os.path.join("./eval", sess_id)
    eval_set = "val" if dataset == "VOC2012" else "test"
    if cues is None or gts is None:
        from PIL import Image

        from .. import keras_store as ks

        st = ks.read_settings(settings)
        img_size = 321 if model_type in ["VGG16", "VGG16bg"] else 224
        ds = ks.Dataset(data_type=dataset, size=img_size, batch_size=batch_size,
                        database_dir=st["DATA_ROOT"] if settings is not None else None)
        eval_set = ds.sets[ds.is_evals.index(True)]
        gen_eval = ds.set_gens[eval_set]
        if cues is None:
            path = os.path.join(eval_dir, "localization_cues_val.pickle")
            if not os.path.exists(path):  # generate first if not already existing (demo.py:398-400)
                gen_cues(dataset, model_type, thresh, batch_size, run_train=False, is_verbose=is_verbose, settings=settings,
                         out_dir=eval_dir)
            with open(path, "rb") as f:
                cues = pickle.load(f, encoding="iso-8859-1")
        if gts is None:
            gt_dir = os.path.join(os.path.dirname(gen_eval.directory), "SegmentationClassAug")
            names = [os.path.splitext(f)[0] + ".png" for f in gen_eval.filenames]
            if dataset == "VOC2012":  # palette PNG: the index IS the class (cv2.imread(...)[:, :, 0] after the colour swap)
                gts = [np.asarray(Image.open(os.path.join(gt_dir, n))) for n in names]
            else:
                gts = [np.asarray(Image.open(os.path.join(gt_dir, n)).convert("RGB")) for n in names]
    if dataset == "VOC2012":
        names_c = list(class_names or VOC_SEG_CLASS_NAMES)
        n, empty = len(names_c), 0
    else:
        from ..step.eval_cam import DEEPGLOBE_CLS_COLOURS

        colours = list(colours or DEEPGLOBE_CLS_COLOURS)  # get_colours(dataset)[:-1]: the 'unknown' class is not scored
        names_c = list(class_names or ["class%d" % k for k in range(len(colours))])
        n, empty = len(colours), len(colours)
    ctx = ctx or default_context()

    def gt_index(g):
        if dataset == "VOC2012":
            return np.where(g < n, g, n).astype(np.uint8)  # 255 (void border) is no class's ground truth (gt_idx == k never holds)
        from ..hsn.demo import gt_index_from_colours

        return gt_index_from_colours(g, colours)

    conf = cue_confusion(ctx, cues, [gt_index(np.asarray(g)) for g in gts], n, empty, batch_size, is_verbose)
    inter = np.diag(conf)[:n].astype(np.float64)
    union = conf[:n, :].sum(1) + conf[:, :n].sum(0) - inter  # (gt == k) | (pred == k) over every pixel
    iou = inter / (union + 1e-7)
    out = {"intersects": inter, "unions": union.astype(np.float64), "IoU": iou, "mIoU": float(np.mean(iou)), "classes": names_c}
    os.makedirs(eval_dir, exist_ok=True)
    base = os.path.join(eval_dir, "metrics_" + sess_id + "_" + eval_set)
    with open(base + ".csv", "w") as f:
        f.write(",Class,IoU\n")
        for k, (c, v) in enumerate(zip(names_c + ["Mean"], list(iou) + [out["mIoU"]])):
            f.write("%d,%s,%r\n" % (k, c, float(v)))
    try:  # the reference writes df.to_excel (demo.py:481-484); pandas needs openpyxl for that
        import pandas as pd

        pd.DataFrame({"Class": names_c + ["Mean"], "IoU": list(iou) + [out["mIoU"]]}, columns=["Class", "IoU"]).to_excel(base + ".xlsx")
    except Exception:
        pass
    return out


def eval_cues_adp(model_type, sess_id, batch_size, size, set_name, should_saveimg=False, is_verbose=True, *, model, alpha,
                  thresholds, images, gts, thresh=0.2, all_classes=None, out_dir=None):
    """02_cues/demo.py:487-640 (ADP): the seeds of gen_cues_adp for both HTT types, then per type and class
    `pred_mask = cv2.resize(cues[:, :, k], (size, size), INTER_NEAREST) == 1` against the colour-coded ground truth:
    intersects, unions, predicted_totals, gt_totals over the set; IoU = I / U (no epsilon: NaN for a class that never occurs,
    as in the reference), 'precision' = I / (gt_totals + 1e-5) and 'recall' = I / (predicted_totals + 1e-5) -- the reference's
    names for these two ratios, kept -- and their means; `metrics_ADP-<htt>_<set>_<model>.csv` (+ `.xlsx` with openpyxl).
    `gts` {'morph': [...], 'func': [...]}: (size, size, 3) RGB ground-truth maps (ADPCues.read_gt_batch).  The one-hot seeds are
    exclusive after the overlap resolution, so the per-class masks are one label map and all four sums come from one confusion
    matrix on the device.  Debug images are not rendered."""
    from ..hsn.demo import ADPClasses, gt_index_from_colours
    from ..step.eval_cam import ADP_CLS_COLOURS

    cues = gen_cues_adp(model_type, thresh, batch_size, size, None, set_name, is_verbose, model=model, alpha=alpha,
                        thresholds=thresholds, images=images, all_classes=all_classes)
    ac = ADPClasses(all_classes)
    eval_dir = out_dir or os.path.join("./eval", sess_id)
    os.makedirs(eval_dir, exist_ok=True)
    out = {}
    for htt in ("morph", "func"):
        colours = [tuple(int(v) for v in c) for c in ADP_CLS_COLOURS[htt]]
        names = ac.classes["valid_" + htt]
        n = len(colours)
        assert n == len(names)
        gidx = [gt_index_from_colours(np.asarray(g), colours) for g in gts[htt]]
        conf = cue_confusion(model.ctx, cues[htt], gidx, n, n, batch_size)
        inter = np.diag(conf)[:n].astype(np.float64)
        gt_tot, pred_tot = conf[:n, :].sum(1).astype(np.float64), conf[:, :n].sum(0).astype(np.float64)
        union = gt_tot + pred_tot - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        prec, rec = inter / (gt_tot + 1e-5), inter / (pred_tot + 1e-5)
        out[htt] = {"intersects": inter, "unions": union, "gt_totals": gt_tot, "predicted_totals": pred_tot, "IoU": iou,
                    "Precision": prec, "Recall": rec, "mIoU": float(np.mean(iou)), "classes": list(names)}
        if is_verbose:
            print("\tmIoU (%s): %s" % (htt, out[htt]["mIoU"]))
        base = os.path.join(eval_dir, "metrics_ADP-" + htt + "_" + str(set_name) + "_" + model_type)
        with open(base + ".csv", "w") as f:
            f.write(",Class,IoU,Precision,Recall\n")
            rows = zip(list(names) + ["Mean"], list(iou) + [np.mean(iou)], list(prec) + [np.mean(prec)], list(rec) + [np.mean(rec)])
            for k, (c, a, b, d) in enumerate(rows):
                f.write("%d,%s,%r,%r,%r\n" % (k, c, float(a), float(b), float(d)))
        try:
            import pandas as pd

            pd.DataFrame({"Class": list(names) + ["Mean"], "IoU": list(iou) + [np.mean(iou)], "Precision": list(prec) + [np.mean(prec)],
                          "Recall": list(rec) + [np.mean(rec)]}, columns=["Class", "IoU", "Precision", "Recall"]).to_excel(base + ".xlsx")
        except Exception:
            pass
    return out, cues
