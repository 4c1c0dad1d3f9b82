"""cam_to_ir_label step on the device -- mirror of 03b_irn/step/cam_to_ir_label.py.

Per image the reference runs pydensecrf twice (confident-foreground and confident-background label maps,
cam_to_ir_label.py:42-58), building the same two lattices both times.  Here the image's lattices are
built once (wsc_crf_create) and the two label-unary mean-field runs share them."""
import os

import numpy as np

from .. import _lib
from ..misc import imutils


def _crf_labels(ctx, crf, labels, n_labels, t=10, gt_prob=0.7):
    """imutils.crf_inference_label on an already built lattice pair."""
    h, w = labels.shape
    U = np.ascontiguousarray(imutils.unary_from_labels(labels, n_labels, gt_prob=gt_prob, zero_unsure=False))
    am_dev = ctx.alloc(h * w * 4)
    crf.inference(ctx.to_device(U), n_labels, 3.0, 10.0, t, None, am_dev)
    return ctx.to_host(am_dev, (h, w), np.int32).astype(np.int64)


def ir_label_voc12(img, cam_dict, conf_fg_thres=0.30, conf_bg_thres=0.05, ctx=None):
    """cam_to_ir_label.py:42-58 for one VOC image: uint8 (H, W) label map, 255 = unreliable region."""
    ctx = ctx or imutils.default_context()
    img = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
    h, w = img.shape[:2]
    keys = np.pad(cam_dict["keys"] + 1, (1, 0), mode="constant")
    n_labels = keys.shape[0]
    if n_labels == 1:  # no foreground class: everything is background
        return np.zeros((h, w), np.uint8)
    crf = _lib.Crf(ctx, ctx.to_device(img), 1, h, w, 3.0, 50.0, 5.0)  # upstream irn crf_inference_label parameters
    try:
        fg_cam = np.pad(cam_dict["high_res"], ((1, 0), (0, 0), (0, 0)), mode="constant", constant_values=conf_fg_thres)
        fg_conf = keys[_crf_labels(ctx, crf, np.argmax(fg_cam, axis=0), n_labels)]
        bg_cam = np.pad(cam_dict["high_res"], ((1, 0), (0, 0), (0, 0)), mode="constant", constant_values=conf_bg_thres)
        bg_conf = keys[_crf_labels(ctx, crf, np.argmax(bg_cam, axis=0), n_labels)]
    finally:
        crf.close()
    conf = fg_conf.copy()
    conf[fg_conf == 0] = 255
    conf[bg_conf + fg_conf == 0] = 0
    return conf.astype(np.uint8)


def _work(process_id, infer_dataset, args):
    """cam_to_ir_label.py:18-95 for the VOC branch: reads <cam_out_dir>/<name>.npy, writes the IR label PNG."""
    from PIL import Image

    databin = infer_dataset[process_id]
    ctx = imutils.default_context(getattr(args, "device", 0))
    for i in range(len(databin)):
        pack = databin[i]
        name = pack["name"]
        cam_dict = np.load(os.path.join(args.cam_out_dir, name + ".npy"), allow_pickle=True).item()
        if len(cam_dict["keys"]) == 0:
            conf = np.zeros(np.asarray(pack["img"]).shape[:2], np.uint8)
        else:
            conf = ir_label_voc12(pack["img"], cam_dict, args.conf_fg_thres, args.conf_bg_thres, ctx=ctx)
        Image.fromarray(conf).save(os.path.join(args.ir_label_out_dir, name + ".png"))


def run(args):
    from ..misc import torchutils

    if args.dataset != "voc12":
        raise KeyError("Dataset %s not yet implemented" % args.dataset)
    os.makedirs(args.ir_label_out_dir, exist_ok=True)
    dataset = torchutils.split_dataset(args.dataset_obj, 1)
    _work(0, dataset, args)
