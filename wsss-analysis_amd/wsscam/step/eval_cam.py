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
