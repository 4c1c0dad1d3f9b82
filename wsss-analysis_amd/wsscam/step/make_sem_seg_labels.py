"""make_sem_seg_labels step -- drop-in for 03b_irn/step/make_sem_seg_labels.py (run :146-175, _work :22-143):
IRNet inference.  For every image: EdgeDisplacement network -> class-boundary map; the strided CAMs that
make_cam wrote are propagated along the boundary-derived affinities (indexing.propagate_to_edge), upsampled to
the image size, max-normalised, thresholded against a constant background score and arg-maxed; the label map is
written as `<sem_seg_out_dir>/<name>.png` plus the colour and overlay renderings in `sem_seg_clr_out_dir`.

Same `args` fields as the reference (`irn_network, model_dir, dataset, tag, num_classes, use_cls,
irn_weights_name, infer_list, norm_mode, outsize, dev_root, split, cam_out_dir, beta, exp_times,
sem_seg_bg_thres, sem_seg_out_dir, sem_seg_clr_out_dir, class_colours, overlay_r`); optional `state_dict`,
`dataset_obj`, `irn_precision`, `n_gpus`, `irn_crop_size` like the make_cam mirror.  The network and the random
walk run on the device; PNG encoding is host I/O (PIL)."""
import importlib
import os

import numpy as np

from .. import _lib
from ..misc import indexing, torchutils


def _build_model(args):
    mod = args.irn_network
    if not mod.startswith("wsscam."):
        mod = "wsscam." + mod
    cls = getattr(importlib.import_module(mod), "EdgeDisplacement")
    kw = {}
    if getattr(args, "irn_crop_size", None):
        kw["crop_size"] = int(args.irn_crop_size)
    if getattr(args, "irn_precision", None) is not None:
        kw["precision"] = args.irn_precision
    try:  # vgg16_irn / m7_irn signature (make_sem_seg_labels.py:147-149)
        return cls(args.model_dir, args.dataset, args.tag, args.num_classes, args.use_cls, **kw)
    except TypeError:  # resnet50_irn.EdgeDisplacement(model_dir, num_classes, crop_size, stride)
        return cls(args.model_dir, args.num_classes, **kw)


def _upsample_norm(ctx, rw, size, crop):
    """F.interpolate(rw, size, bilinear, align_corners=False)[..., 0, :crop[0], :crop[1]] / max -- on the device."""
    K, _, h, w = rw.shape
    src = ctx.to_device(np.ascontiguousarray(rw.reshape(K, h, w)))
    dst = ctx.alloc(K * size[0] * size[1] * 4)
    _lib.bilinear_resize(ctx, src, K, h, w, dst, int(size[0]), int(size[1]))
    up = ctx.to_host(dst, (K, int(size[0]), int(size[1])), np.float32)[:, :crop[0], :crop[1]]
    src.free()
    dst.free()
    with np.errstate(invalid="ignore", divide="ignore"):  # an all-zero map gives NaN, as rw_up / torch.max(rw_up) does
        return up / np.max(up)


def sem_seg_one(model, pack, cam_dict, args):
    """One image: (2,3,h,w) input pair + its make_cam dict -> label map (make_sem_seg_labels.py:34-104)."""
    ctx = model.ctx
    size = np.asarray(pack["size"]).reshape(-1)
    edge, _dp = model.forward(np.asarray(pack["img"], dtype=np.float32))  # edge (1,fh,fw)
    cams = np.asarray(cam_dict["cam"], dtype=np.float32)
    keys_in = np.asarray(cam_dict["keys"])

    def fit_edge(target_hw):
        if edge.shape[1:] == tuple(target_hw):
            return edge
        src = ctx.to_device(np.ascontiguousarray(edge))
        dst = ctx.alloc(int(target_hw[0]) * int(target_hw[1]) * 4)
        _lib.bilinear_resize(ctx, src, 1, edge.shape[1], edge.shape[2], dst, int(target_hw[0]), int(target_hw[1]))
        out = ctx.to_host(dst, (1, int(target_hw[0]), int(target_hw[1])), np.float32)
        src.free()
        dst.free()
        return out

    if args.dataset == "voc12":
        if len(keys_in) == 0:
            return np.zeros(tuple(size), dtype="uint8")
        keys = np.pad(keys_in + 1, (1, 0), mode="constant")
        e = fit_edge(cams.shape[1:])
        rw = indexing.propagate_to_edge(cams, e, beta=args.beta, exp_times=args.exp_times, radius=5, ctx=ctx)
        rw_up = _upsample_norm(ctx, rw, size, size)
        rw_up_bg = np.concatenate((np.full((1,) + rw_up.shape[1:], float(args.sem_seg_bg_thres), np.float32), rw_up))
        return keys[np.argmax(rw_up_bg, axis=0)]
    if args.dataset in ("adp_morph", "adp_func"):
        e = fit_edge(cams.shape[1:])
        rw = indexing.propagate_to_edge(cams, e, beta=args.beta, exp_times=args.exp_times, radius=5, ctx=ctx)
        rw_up = _upsample_norm(ctx, rw, size, size)
        return keys_in[np.argmax(rw_up, axis=0)]
    if args.dataset in ("deepglobe", "deepglobe_balanced"):
        if len(keys_in) == 0:
            return 5 * np.ones(tuple(size // 4))
        down_fac = 6
        small = [v // down_fac for v in cams.shape[1:]]
        src = ctx.to_device(np.ascontiguousarray(cams))
        dst = ctx.alloc(cams.shape[0] * small[0] * small[1] * 4)
        _lib.bilinear_resize(ctx, src, cams.shape[0], cams.shape[1], cams.shape[2], dst, small[0], small[1])
        cams_s = ctx.to_host(dst, (cams.shape[0], small[0], small[1]), np.float32)
        src.free()
        dst.free()
        e = fit_edge(small)
        rw = indexing.propagate_to_edge(cams_s, e, beta=args.beta, exp_times=args.exp_times, radius=5, ctx=ctx)
        rw_up = _upsample_norm(ctx, rw, size // 4, size // 4)
        return keys_in[np.argmax(rw_up, axis=0)]
    raise KeyError("Dataset %s not yet implemented" % args.dataset)


def _save(args, name, rw_pred, orig_rgb=None):
    """label PNG + colour PNG (+ overlay when the original image is available): make_sem_seg_labels.py:106-127."""
    from PIL import Image

    Image.fromarray(rw_pred.astype(np.uint8)).save(os.path.join(args.sem_seg_out_dir, name + ".png"))
    colours = getattr(args, "class_colours", None)
    clr_dir = getattr(args, "sem_seg_clr_out_dir", None)
    if colours is None or clr_dir is None:
        return
    clr = np.zeros(list(rw_pred.shape) + [3], dtype=np.uint8)
    off = 0
    for t in ("bg", "fg"):
        for i, c in enumerate(colours[t]):
            for ch in range(3):
                clr[:, :, ch] += np.uint8(c[ch]) * np.uint8(rw_pred == (i + off))
        off += len(colours[t])
    Image.fromarray(clr).save(os.path.join(clr_dir, name + ".png"))
    if orig_rgb is not None:
        if orig_rgb.shape[:2] != clr.shape[:2]:
            orig_rgb = np.asarray(Image.fromarray(orig_rgb).resize((clr.shape[1], clr.shape[0]), Image.BILINEAR))
        r = float(getattr(args, "overlay_r", 0.75))
        over = np.uint8((1 - r) * np.float32(orig_rgb) + r * np.float32(clr))
        Image.fromarray(over).save(os.path.join(clr_dir, name + "_overlay.png"))


def _work(process_id, model, dataset, args):
    databin = dataset[process_id]
    model.cuda(process_id)
    for i in range(len(databin)):
        pack = databin[i]
        name = pack["name"]
        cam_dict = np.load(os.path.join(args.cam_out_dir, name + ".npy"), allow_pickle=True).item()
        pred = sem_seg_one(model, pack, cam_dict, args)
        orig = pack.get("orig_img")
        _save(args, name, pred, None if orig is None else np.asarray(orig)[0])
    model.ctx.sync()


def build_dataset(args):
    if getattr(args, "dataset_obj", None) is not None:
        return args.dataset_obj
    from . import make_cam

    shim = type("A", (), dict(vars(args)))()
    shim.val_list = args.infer_list
    shim.cam_scales = (1.0,)
    return make_cam.build_dataset(shim)


def run(args):
    """03b_irn/step/make_sem_seg_labels.py:146-175."""
    model = _build_model(args)
    if getattr(args, "state_dict", None) is not None:
        model.load_state_dict(args.state_dict, strict=False)
    else:
        import torch

        model.load_state_dict(torch.load(args.irn_weights_name, map_location="cpu"), strict=False)
    model.eval()
    n_gpus = int(getattr(args, "n_gpus", 0)) or __import__("torch").cuda.device_count()
    if n_gpus < 1:
        raise _lib.WscError(_lib.WSC_ERR_NO_DEVICE, "make_sem_seg_labels needs at least one gfx950 device")
    os.makedirs(args.sem_seg_out_dir, exist_ok=True)
    if getattr(args, "sem_seg_clr_out_dir", None):
        os.makedirs(args.sem_seg_clr_out_dir, exist_ok=True)
    dataset = torchutils.split_dataset(build_dataset(args), n_gpus)
    if n_gpus == 1:
        _work(0, model, dataset, args)
    else:
        import torch.multiprocessing as mp

        mp.spawn(_work, nprocs=n_gpus, args=(model, dataset, args), join=True)
