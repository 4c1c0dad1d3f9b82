"""eval_sem_seg step -- mirror of 03b_irn/step/eval_sem_seg.py (run :12-64): reads the label PNGs that
make_sem_seg_labels wrote, resizes them (nearest) to the evaluation size where the dataset asks for it,
accumulates the semantic-segmentation confusion matrix and writes `<run_name>_<split>_iou.csv` + the two
`[eval_sem_seg, <split>]` log lines.  Host-only bookkeeping (no device work).

chainercv is not in this image: `args.gt_labels` (dict name -> int label map, -1 / 255 = ignore) and `args.ids`
stand in for VOCSemanticSegmentationDataset; the confusion matrix follows chainercv's
calc_semantic_segmentation_confusion (bincount of n_class * gt + pred over the pixels with gt >= 0, growing
n_class to the largest label seen)."""
import os

import numpy as np


def calc_semantic_segmentation_confusion(pred_labels, gt_labels):
    n_class = 0
    confusion = np.zeros((0, 0), dtype=np.int64)
    for pred, gt in zip(pred_labels, gt_labels):
        pred = np.asarray(pred).astype(np.int64).ravel()
        gt = np.asarray(gt).astype(np.int64).ravel()
        if pred.shape != gt.shape:
            raise ValueError("Shape of ground truth and prediction should be same.")
        lb_max = int(max(pred.max(), gt.max()))
        if lb_max >= n_class:
            grown = np.zeros((lb_max + 1, lb_max + 1), dtype=np.int64)
            grown[:n_class, :n_class] = confusion
            n_class, confusion = lb_max + 1, grown
        mask = gt >= 0
        confusion += np.bincount(n_class * gt[mask] + pred[mask], minlength=n_class ** 2).reshape(n_class, n_class)
    return confusion


def _resize_nearest(lab, outsize):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src index = floor(dst * scale)."""
    H, W = lab.shape
    ys = np.minimum((np.arange(outsize[1]) * (H / outsize[1])).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(outsize[0]) * (W / outsize[0])).astype(np.int64), W - 1)
    return lab[ys][:, xs]


def run(args):
    from PIL import Image

    if args.dataset == "voc12":
        outsize = None
    elif args.dataset in ("adp_morph", "adp_func"):
        outsize = (1088, 1088)
    elif args.dataset in ("deepglobe", "deepglobe_balanced"):
        outsize = (2448, 2448)
    else:
        raise KeyError("Dataset %s not yet implemented" % args.dataset)
    ids = list(args.ids)
    labels, preds = [], []
    for name in ids:
        gt = np.asarray(args.gt_labels[name]).astype(np.int64)
        gt[gt == 255] = -1  # chainercv's VOC loader marks 'ignore' as -1
        labels.append(gt)
        cls = np.asarray(Image.open(os.path.join(args.sem_seg_out_dir, name + ".png"))).astype(np.uint8)
        cls[cls == 255] = 0
        if outsize is not None:
            cls = _resize_nearest(cls, outsize)
        preds.append(cls.copy())
    confusion = calc_semantic_segmentation_confusion(preds, labels)
    gtj = confusion.sum(axis=1)
    resj = confusion.sum(axis=0)
    gtjresj = np.diag(confusion)
    with np.errstate(divide="ignore", invalid="ignore"):
        denominator = gtj + resj - gtjresj
        iou = gtjresj / denominator
    miou = np.array([np.nanmean(iou)])
    if args.dataset in ("deepglobe", "deepglobe_balanced"):
        row_names = args.class_names["bg"] + args.class_names["fg"][:-1] + ["miou"]
    else:
        row_names = args.class_names["bg"] + args.class_names["fg"] + ["miou"]
    data = np.concatenate((iou, miou), axis=0)
    os.makedirs(args.eval_dir, exist_ok=True)
    with open(os.path.join(args.eval_dir, args.run_name + "_" + args.split + "_iou.csv"), "w") as f:
        f.write(",iou\n")  # pandas.DataFrame(data, index=row_names, columns=['iou']).to_csv(index=True)
        for n, v in zip(row_names, data):
            f.write("%s,%s\n" % (n, repr(float(v)) if not np.isnan(v) else ""))
    with open(args.logfile, "a") as f:
        f.write("[eval_sem_seg, " + args.split + "] iou: " + str(list(iou)) + "\n")
        f.write("[eval_sem_seg, " + args.split + "] miou: " + str(miou[0]) + "\n")
    return {"confusion": confusion, "iou": iou, "miou": float(miou[0])}
