"""eval_cam step on the device -- mirror of 03b_irn/step/eval_cam.py.

The reference reloads every `<name>.npy`, pads a constant background channel (cam_eval_thres), takes the
arg-max, maps it through `keys` and accumulates chainercv's confusion matrix on the host.  Here a batch's
high_res maps are consumed where wsc_cam_postprocess left them (no .npy round trip needed when only the
mIoU is wanted) by wsc_cam_eval_confusion; the IoU / precision / recall table, the CSV and the
`[eval_cam, split] miou:` log line keep the reference's formats (eval_cam.py:89-115)."""
import os

import numpy as np

from .. import _lib


def scores_from_confusion(confusion):
    """eval_cam.py:89-98: per-class iou / precision / recall and their nan-means."""
    confusion = np.asarray(confusion, dtype=np.float64)
    gtj = confusion.sum(axis=1)
    resj = confusion.sum(axis=0)
    gtjresj = np.diag(confusion)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = gtjresj / (gtj + resj - gtjresj)
        precision = gtjresj / resj
        recall = gtjresj / gtj
    return {"iou": iou, "precision": precision, "recall": recall, "miou": np.nanmean(iou),
            "mprecision": np.nanmean(precision), "mrecall": np.nanmean(recall)}


class ConfusionAccumulator:
    """Device-resident confusion matrix carried over the batches of a dataset."""

    def __init__(self, ctx, n_class=21, cam_eval_thres=0.15, ignore_label=255):
        self.ctx, self.n_class, self.thres, self.ignore = ctx, n_class, cam_eval_thres, ignore_label
        self.conf_dev = ctx.alloc(n_class * n_class * 8)
        _lib.check(ctx._lib.wsc_memset(ctx.h, self.conf_dev.ptr, 0, n_class * n_class * 8))

    def add_batch(self, highres_dev, sizes, keys_per_image, highres_off, gt_labels, want_pred=False):
        """gt_labels: list of uint8 (H0, W0) arrays (255 = ignore).  Returns predicted label maps if asked."""
        gt = np.concatenate([np.ascontiguousarray(g, dtype=np.uint8).ravel() for g in gt_labels])
        gt_dev = self.ctx.to_device(gt)
        pred_dev = self.ctx.alloc(gt.size) if want_pred else None
        _lib.cam_eval_confusion(self.ctx, highres_dev, sizes, keys_per_image, highres_off, self.thres, gt_dev,
                                self.n_class, self.conf_dev, pred_dev, self.ignore)
        if not want_pred:
            return None
        flat = self.ctx.to_host(pred_dev, (gt.size,), np.uint8)
        out, o = [], 0
        for (H0, W0) in sizes:
            out.append(flat[o:o + H0 * W0].reshape(H0, W0))
            o += H0 * W0
        return out

    def confusion(self):
        return self.ctx.to_host(self.conf_dev, (self.n_class, self.n_class), np.int64)


def write_report(args, confusion, row_names):
    """CSV + log lines exactly as eval_cam.py:100-115 writes them (pandas is optional here)."""
    s = scores_from_confusion(confusion)
    data = np.column_stack((np.append(s["iou"], s["miou"]), np.append(s["precision"], s["mprecision"]),
                            np.append(s["recall"], s["mrecall"])))
    path = os.path.join(args.eval_dir, args.run_name + "_" + args.split + "_cam_iou.csv")
    with open(path, "w") as f:
        f.write(",iou,precision,recall\n")
        for name, row in zip(list(row_names) + ["mean"], data):
            f.write("%s,%s\n" % (name, ",".join(repr(float(v)) for v in row)))
    with open(args.logfile, "a") as f:
        f.write("[eval_cam, " + args.split + "] iou: " + str(list(s["iou"])) + "\n")
        f.write("[eval_cam, " + args.split + "] miou: " + str(s["miou"]) + "\n")
    return s


class VOCSegLabels:
    """Ground truth of the VOC segmentation split the reference reads through chainercv's
    VOCSemanticSegmentationDataset (eval_cam.py:21-22,34): ids from ImageSets/Segmentation/<split>.txt, label maps
    from SegmentationClass/<id>.png (palette indices; 255 = ignore, which chainercv maps to -1)."""

    def __init__(self, split, data_dir):
        with open(os.path.join(data_dir, "ImageSets", "Segmentation", split + ".txt")) as f:
            self.ids = [l.strip() for l in f if l.strip()]
        self.data_dir = data_dir

    def __len__(self):
        return len(self.ids)

    def label(self, i):
        from PIL import Image

        return np.asarray(Image.open(os.path.join(self.data_dir, "SegmentationClass", self.ids[i] + ".png")), dtype=np.uint8)


def _colour_maps(args, cls_labels):
    """eval_cam.py:67-75: class colours in bg-then-fg order."""
    clr = np.zeros(cls_labels.shape + (3,), dtype=np.uint8)
    off = 0
    for t in ("bg", "fg"):
        for i, c in enumerate(args.class_colours[t]):
            clr[cls_labels == (i + off)] = np.asarray(c, dtype=np.uint8)
        off += len(args.class_colours[t])
    return clr


def run(args, ctx=None, batch_images=32):
    """03b_irn/step/eval_cam.py:19-115 for dataset == 'voc12': every `<cam_out_dir>/<id>.npy` -> arg-max over
    [cam_eval_thres | high_res] -> keys -> confusion against the ground truth, IoU / precision / recall CSV and the
    `[eval_cam, split] miou:` log line.  The arg-max and the confusion matrix run on the device
    (wsc_cam_eval_confusion); label PNGs / colour PNGs / overlays are written when args.cam_clr_out_dir is set.
    `args.seg_labels` may hand in an object with `.ids` and `.label(i)` (default: VOCSegLabels on args.dev_root)."""
    if args.dataset != "voc12":
        raise NotImplementedError("eval_cam.run: only the voc12 branch runs on the device (ADP / DeepGlobe evaluate "
                                  "after a nearest-neighbour resize to 1088^2 / 2448^2: eval_cam.py:24-31,62-63)")
    labels = getattr(args, "seg_labels", None) or VOCSegLabels(args.chainer_eval_set, args.dev_root)
    own_ctx = ctx is None
    if own_ctx:
        ctx = _lib.Context(int(getattr(args, "device", 0)))
    n_class = len(args.class_names["bg"]) + len(args.class_names["fg"])
    acc = ConfusionAccumulator(ctx, n_class=n_class, cam_eval_thres=args.cam_eval_thres)
    clr_dir = getattr(args, "cam_clr_out_dir", None)
    if clr_dir:
        os.makedirs(clr_dir, exist_ok=True)
    ids = list(labels.ids)
    for i0 in range(0, len(ids), batch_images):
        chunk = range(i0, min(i0 + batch_images, len(ids)))
        maps, sizes, keys, offs, gts = [], [], [], [], []
        tot = 0
        for i in chunk:
            gt = np.asarray(labels.label(i))
            cam_dict = np.load(os.path.join(args.cam_out_dir, ids[i] + ".npy"), allow_pickle=True).item()
            k = np.asarray(cam_dict["keys"], dtype=np.int64)
            hr = np.asarray(cam_dict["high_res"], dtype=np.float32).reshape((len(k),) + gt.shape)
            maps.append(hr.ravel())
            sizes.append(gt.shape)
            keys.append(k)
            offs.append(tot)
            tot += hr.size
            gts.append(gt)
        h_dev = ctx.to_device(np.concatenate(maps) if tot else np.zeros(1, np.float32))
        preds = acc.add_batch(h_dev, sizes, keys, np.asarray(offs, np.int64), gts, want_pred=bool(clr_dir))
        if clr_dir:
            from PIL import Image

            for i, pred in zip(chunk, preds):
                clr = _colour_maps(args, pred)
                Image.fromarray(clr).save(os.path.join(clr_dir, ids[i] + ".png"))  # eval_cam.py:76 (overwrites the label PNG of :66)
                img_path = getattr(args, "img_path_of", None)
                if img_path is not None:
                    orig = np.asarray(Image.open(img_path(ids[i])).convert("RGB"))
                    over = np.uint8((1 - args.overlay_r) * np.float32(orig) + args.overlay_r * np.float32(clr))
                    Image.fromarray(over).save(os.path.join(clr_dir, ids[i] + "_overlay.png"))
    conf = acc.confusion()
    os.makedirs(args.eval_dir, exist_ok=True)
    s = write_report(args, conf, list(args.class_names["bg"]) + list(args.class_names["fg"]))
    if own_ctx:
        ctx.close()
    return conf, s
