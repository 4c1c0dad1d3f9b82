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


# class colours of the ground-truth PNGs (dataset constants: adp/adp_semantic_segmentation_dataset.py:10-17,
# deepglobe/deepglobe_semantic_segmentation_dataset.py:10-11)
ADP_CLS_COLOURS = {
    "morph": [(255, 255, 255), (0, 0, 128), (0, 128, 0), (255, 165, 0), (255, 192, 203), (255, 0, 0), (173, 20, 87),
              (176, 141, 105), (3, 155, 229), (158, 105, 175), (216, 27, 96), (244, 81, 30), (124, 179, 66), (142, 36, 255),
              (240, 147, 0), (204, 25, 165), (121, 85, 72), (142, 36, 170), (179, 157, 219), (121, 134, 203), (97, 97, 97),
              (167, 155, 142), (228, 196, 136), (213, 0, 0), (4, 58, 236), (0, 150, 136), (228, 196, 65), (239, 108, 0),
              (74, 21, 209)],
    "func": [(255, 255, 255), (3, 155, 229), (0, 0, 128), (0, 128, 0), (173, 20, 87)],
}
DEEPGLOBE_CLS_COLOURS = [(0, 255, 255), (255, 255, 0), (255, 0, 255), (0, 255, 0), (0, 0, 255), (255, 255, 255)]


def label_from_colours(rgb, colours):
    """_read_label of both dataset classes (:55-62 / :49-56): label += (pixel == colour i) * i over the colour table; a pixel
    of no listed colour stays 0."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    label = np.zeros(rgb.shape[:2], dtype=np.int32)
    for i, c in enumerate(colours):
        label += np.all(rgb == np.asarray(c, dtype=np.uint8)[None, None, :], axis=2) * i
    return label


class _ColourSegLabels:
    def __init__(self, data_dir, split_file, label_dir, colours):
        with open(os.path.join(data_dir, "ImageSets", "Segmentation", split_file + ".txt")) as f:
            self.ids = [l.strip() for l in f if l.strip()]
        self.label_dir, self.colours = label_dir, colours

    def __len__(self):
        return len(self.ids)

    def label(self, i):
        from PIL import Image

        rgb = np.asarray(Image.open(os.path.join(self.label_dir, self.ids[i] + ".png")).convert("RGB"))
        return label_from_colours(rgb, self.colours).astype(np.uint8)


class ADPSegLabels(_ColourSegLabels):
    """adp/adp_semantic_segmentation_dataset.py:19-70: ids of ImageSets/Segmentation/<split>.txt ('evaluation' reads
    'segtest'), labels decoded from SegmentationClassAug/ADP-<htt>/<id>.png by colour."""

    def __init__(self, split, data_dir, htt_type):
        if htt_type not in ("morph", "func"):
            raise ValueError("please pick HTT type from 'morph', 'func'")
        if split not in ("train", "tuning", "evaluation"):
            raise ValueError("please pick split from 'train', 'tuning', 'evaluation'")
        super().__init__(data_dir, "segtest" if split == "evaluation" else split,
                         os.path.join(data_dir, "SegmentationClassAug", "ADP-" + htt_type), ADP_CLS_COLOURS[htt_type])


class DeepGlobeSegLabels(_ColourSegLabels):
    """deepglobe/deepglobe_semantic_segmentation_dataset.py:13-64: 'train' -> train75 / train37.5 (balanced), 'test'."""

    def __init__(self, split, data_dir, is_balanced=False):
        if split not in ("train", "test"):
            raise ValueError("please pick split from 'train', 'test'")
        name = "test" if split == "test" else ("train37.5" if is_balanced else "train75")
        super().__init__(data_dir, name, os.path.join(data_dir, "SegmentationClassAug"), DEEPGLOBE_CLS_COLOURS)


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
    """03b_irn/step/eval_cam.py:19-115.  voc12: every `<cam_out_dir>/<id>.npy` -> arg-max over [cam_eval_thres | high_res] ->
    keys -> confusion against the ground truth, IoU / precision / recall CSV and the `[eval_cam, split] miou:` log line.
    ADP / DeepGlobe (:53-63): keys[argmax(high_res | cam)] without a background channel, nearest-neighbour resize to the
    ground truth's size (1088^2 / 2448^2 in the reference), same report (DeepGlobe without its last class, :104-105).  The arg-max and the confusion matrix run on the device
    (wsc_cam_eval_confusion); label PNGs / colour PNGs / overlays are written when args.cam_clr_out_dir is set.
    `args.seg_labels` may hand in an object with `.ids` and `.label(i)` (default: VOCSegLabels on args.dev_root)."""
    voc = args.dataset == "voc12"
    adp = args.dataset in ("adp_morph", "adp_func")
    dg = args.dataset in ("deepglobe", "deepglobe_balanced")
    if not (voc or adp or dg):
        raise KeyError("Dataset %s not yet implemented" % args.dataset)
    labels = getattr(args, "seg_labels", None)
    if labels is None:
        if voc:
            labels = VOCSegLabels(args.chainer_eval_set, args.dev_root)
        elif adp:
            labels = ADPSegLabels(args.chainer_eval_set, args.dev_root, args.dataset.split("_")[-1])
        else:
            labels = DeepGlobeSegLabels(args.chainer_eval_set, args.dev_root, is_balanced=args.dataset == "deepglobe_balanced")
    if not voc:
        return _run_resized(args, labels, ctx, batch_images, "high_res" if adp else "cam", dg)
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


def _run_resized(args, labels, ctx, batch_images, field, drop_last_class):
    """The ADP / DeepGlobe branch of run(): maps at their own size, prediction resized (cv2 INTER_NEAREST rule) to the ground
    truth's size on the device (wsc_cam_eval_confusion_nn)."""
    own_ctx = ctx is None
    if own_ctx:
        ctx = _lib.Context(int(getattr(args, "device", 0)))
    n_class = len(args.class_names["bg"]) + len(args.class_names["fg"])
    conf_dev = ctx.alloc(n_class * n_class * 8)
    _lib.check(ctx._lib.wsc_memset(ctx.h, conf_dev.ptr, 0, n_class * n_class * 8))
    clr_dir = getattr(args, "cam_clr_out_dir", None)
    if clr_dir:
        os.makedirs(clr_dir, exist_ok=True)
    ids = list(labels.ids)
    # 2448^2 ground truths are 6 MB each: keep the batches small enough for the packed uint8 arrays
    for i0 in range(0, len(ids), batch_images):
        chunk = range(i0, min(i0 + batch_images, len(ids)))
        maps, src, out, keys, offs, gts = [], [], [], [], [], []
        tot = 0
        for i in chunk:
            gt = np.asarray(labels.label(i), dtype=np.uint8)
            cam_dict = np.load(os.path.join(args.cam_out_dir, ids[i] + ".npy"), allow_pickle=True).item()
            k = np.asarray(cam_dict["keys"], dtype=np.int64)
            m = np.asarray(cam_dict[field], dtype=np.float32)
            if m.ndim != 3 or m.shape[0] != len(k) or len(k) == 0:
                raise ValueError("eval_cam: %s has %s maps %s for %d keys (np.argmax of an empty stack fails in the reference too)"
                                 % (ids[i], field, m.shape, len(k)))
            maps.append(m.ravel())
            src.append(m.shape[1:])
            out.append(gt.shape)
            keys.append(k)
            offs.append(tot)
            tot += m.size
            gts.append(gt.ravel())
        m_dev = ctx.to_device(np.concatenate(maps))
        gt_dev = ctx.to_device(np.concatenate(gts))
        pred_dev = ctx.alloc(sum(g.size for g in gts)) if clr_dir else None
        _lib.cam_eval_confusion_nn(ctx, m_dev, src, out, keys, np.asarray(offs, np.int64), gt_dev, n_class, conf_dev, pred_dev)
        if clr_dir:
            from PIL import Image

            from ..voc12.dataloader import resize_bilinear_u8

            flat = ctx.to_host(pred_dev, (sum(g.size for g in gts),), np.uint8)
            o = 0
            for i, shp in zip(chunk, out):
                pred = flat[o:o + shp[0] * shp[1]].reshape(shp)
                o += shp[0] * shp[1]
                clr = _colour_maps(args, pred)
                Image.fromarray(clr).save(os.path.join(clr_dir, ids[i] + ".png"))
                img_path = getattr(args, "img_path_of", None)
                if img_path is not None:
                    orig = np.asarray(Image.open(img_path(ids[i])).convert("RGB"))

                    def _bil(a, hw):  # cv2.resize(..., INTER_LINEAR) on uint8: OpenCV's 8-bit fixed-point rule
                        return a if a.shape[:2] == tuple(hw) else resize_bilinear_u8(a, hw)

                    if drop_last_class:   # DeepGlobe (:79): the image is resized to the prediction
                        orig = _bil(orig, clr.shape[:2])
                    else:                 # ADP (:80-81): the colour map is resized to the image
                        clr = _bil(clr, orig.shape[:2])
                    over = np.uint8((1 - args.overlay_r) * np.float32(orig) + args.overlay_r * np.float32(clr))
                    Image.fromarray(over).save(os.path.join(clr_dir, ids[i] + "_overlay.png"))
    conf = ctx.to_host(conf_dev, (n_class, n_class), np.int64)
    rows = list(args.class_names["bg"]) + list(args.class_names["fg"])
    if drop_last_class:  # eval_cam.py:104-105: the report has no row for the last fg class ('unknown')
        if conf[-1].sum() or conf[:, -1].sum():
            raise ValueError("eval_cam: %d pixels of the dropped last class (%s) take part in the confusion matrix"
                             % (int(conf[-1].sum() + conf[:, -1].sum()), rows[-1]))
        conf, rows = conf[:-1, :-1], rows[:-1]
    os.makedirs(args.eval_dir, exist_ok=True)
    s = write_report(args, conf, rows)
    if own_ctx:
        ctx.close()
    return conf, s
