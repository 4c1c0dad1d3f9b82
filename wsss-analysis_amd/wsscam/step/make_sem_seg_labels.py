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


def _resize(ctx, maps, hw):
    """F.interpolate(maps[None], size=hw, mode='bilinear', align_corners=False)[0] on the device."""
    maps = np.ascontiguousarray(maps, dtype=np.float32)
    if maps.shape[1:] == tuple(hw):
        return maps
    src = ctx.to_device(maps)
    dst = ctx.alloc(maps.shape[0] * int(hw[0]) * int(hw[1]) * 4)
    _lib.bilinear_resize(ctx, src, maps.shape[0], maps.shape[1], maps.shape[2], dst, int(hw[0]), int(hw[1]))
    out = ctx.to_host(dst, (maps.shape[0], int(hw[0]), int(hw[1])), np.float32)
    src.free()
    dst.free()
    return out


def _prepare(ctx, edge, pack, cam_dict, args):
    """Per-image inputs of the random walk (make_sem_seg_labels.py:50-93): (cams, edge at the cam size, keys,
    output size) or a finished label map when there is nothing to propagate."""
    size = np.asarray(pack["size"]).reshape(-1)
    cams = np.asarray(cam_dict["cam"], dtype=np.float32)
    keys_in = np.asarray(cam_dict["keys"])
    if args.dataset == "voc12":
        if len(keys_in) == 0:
            return np.zeros(tuple(size), dtype="uint8")
        return cams, _resize(ctx, edge, cams.shape[1:]), np.pad(keys_in + 1, (1, 0), mode="constant"), size
    if args.dataset in ("adp_morph", "adp_func"):
        return cams, _resize(ctx, edge, cams.shape[1:]), keys_in, size
    if args.dataset in ("deepglobe", "deepglobe_balanced"):
        if len(keys_in) == 0:
            return 5 * np.ones(tuple(size // 4))
        small = [v // 6 for v in cams.shape[1:]]  # down_fac = 6
        return _resize(ctx, cams, small), _resize(ctx, edge, small), keys_in, size // 4
    raise KeyError("Dataset %s not yet implemented" % args.dataset)


def _finish(ctx, rw, keys, size, args):
    rw_up = _upsample_norm(ctx, rw, size, size)
    if args.dataset == "voc12":
        rw_up = np.concatenate((np.full((1,) + rw_up.shape[1:], float(args.sem_seg_bg_thres), np.float32), rw_up))
    return keys[np.argmax(rw_up, axis=0)]


def sem_seg_batch(model, packs, cam_dicts, args):
    """A list of images -> list of label maps: one EdgeDisplacement pass and ONE random-walk pass for all of them
    (every stencil step is a single launch over the whole list), then the per-image upsample / argmax."""
    ctx = model.ctx
    edges, _dp = model.forward_batch(np.stack([np.asarray(p["img"], dtype=np.float32) for p in packs]))
    prepared = [_prepare(ctx, edges[i], p, c, args) for i, (p, c) in enumerate(zip(packs, cam_dicts))]
    todo = [i for i, v in enumerate(prepared) if isinstance(v, tuple)]
    out = [None if isinstance(v, tuple) else v for v in prepared]
    if todo:
        rws = indexing.propagate_to_edge_batch([prepared[i][0] for i in todo], [prepared[i][1] for i in todo], radius=5,
                                               beta=args.beta, exp_times=args.exp_times, ctx=ctx)
        for i, rw in zip(todo, rws):
            out[i] = _finish(ctx, rw, prepared[i][2], prepared[i][3], args)
    return out


def sem_seg_one(model, pack, cam_dict, args):
    """One image: (2,3,h,w) input pair + its make_cam dict -> label map (make_sem_seg_labels.py:34-104)."""
    return sem_seg_batch(model, [pack], [cam_dict], args)[0]


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
    bs = int(getattr(args, "irn_batch_images", 16))  # images per device pass (the reference does one at a time)
    for i0 in range(0, len(databin), bs):
        packs = [databin[i] for i in range(i0, min(i0 + bs, len(databin)))]
        cams = [np.load(os.path.join(args.cam_out_dir, p["name"] + ".npy"), allow_pickle=True).item() for p in packs]
        for p, pred in zip(packs, sem_seg_batch(model, packs, cams, args)):
            orig = p.get("orig_img")
            _save(args, p["name"], pred, None if orig is None else np.asarray(orig)[0])
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
