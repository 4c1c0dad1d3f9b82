"""cam_to_ir_label step on the device -- mirror of 03b_irn/step/cam_to_ir_label.py.

Per image the reference runs pydensecrf once (ADP, DeepGlobe) or twice (VOC: confident-foreground and
confident-background label maps, cam_to_ir_label.py:42-58), building the same two lattices both times, on
cpu_count/2 worker processes.  Here:
  * the images of a device batch go through ONE ragged CRF object as they come from the dataset -- every image its own size
    and its own class count (wsc_crf_v; lattices built once per batch and shared by the VOC branch's two mean-field runs);
    `ir_label_batch` is the same for a batch of one size and one class count (one wsc_crf);
  * the arg-max over [threshold | high_res], the label unaries, the key lookup and the fg/bg merge run on the device
    (wsc_label_unary_from_cam, wsc_crf_inference, wsc_ir_label_combine): per image the host sees the uint8 label map;
  * the dataset is sharded images[g::G] over args.num_workers GPU workers like every other step (run() below).
Outputs as in the reference: <ir_label_out_dir>/<name>.png, and -- when args.ir_label_clr_out_dir is set -- the colour
map and its overlay on the image (cam_to_ir_label.py:77-93).
"""
import os

import numpy as np

from .. import _lib
from ..misc import imutils

CRF_PARAMS = (3.0, 3.0, 50.0, 5.0, 10.0, 10)  # upstream irn crf_inference_label: g_sxy, g_compat, bi_sxy, bi_srgb, bi_compat, t
GT_PROB = 0.7


def _nearest_resize_cv2(img, out_wh):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR is what :61 calls on the uint8 DeepGlobe image (the name is
    historical): OpenCV's 8-bit fixed-point result, voc12.dataloader.resize_bilinear_u8 (an exact /2 decimation is its
    INTER_AREA shortcut; restated from the published algorithm, cv2 absent offline -- SURVEY Q10)."""
    from ..voc12.dataloader import resize_bilinear_u8

    w, h = int(out_wh[0]), int(out_wh[1])
    return resize_bilinear_u8(img, (h, w))


def ir_label_batch(ctx, imgs, maps, keys_list, mode, conf_fg_thres, conf_bg_thres):
    """B images of one size (H, W, 3) uint8 with K class maps each (B, K, H, W) -> uint8 conf maps (B, H, W).
    mode 'voc12': keys = pad(keys + 1, 0), two CRF runs (:42-58); 'fg': keys = [-1] + keys, one run (:27-40, :60-73)."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    B, H, W = imgs.shape[:3]
    K = maps.shape[1]
    N = H * W
    if mode == "voc12":
        keys = np.stack([np.pad(np.asarray(k, dtype=np.int64) + 1, (1, 0), mode="constant") for k in keys_list])
    else:
        keys = np.stack([np.concatenate((np.array([-1]), np.asarray(k, dtype=np.int64))) for k in keys_list])
    g_sxy, g_compat, bi_sxy, bi_srgb, bi_compat, t = CRF_PARAMS
    hr_dev = ctx.to_device(np.ascontiguousarray(maps, dtype=np.float32), pooled=True)
    crf = _lib.Crf(ctx, ctx.to_device(imgs, pooled=True), B, H, W, g_sxy, bi_sxy, bi_srgb)
    try:
        u_dev = ctx.alloc(B * (K + 1) * N * 4, pooled=True)
        fg_dev = ctx.alloc(B * N * 4, pooled=True)
        _lib.label_unary_from_cam(ctx, hr_dev, B, K, N, conf_fg_thres, GT_PROB, u_dev)
        crf.inference(u_dev, K + 1, g_compat, bi_compat, t, None, fg_dev)
        bg_dev = None
        if mode == "voc12":
            bg_dev = ctx.alloc(B * N * 4, pooled=True)
            _lib.label_unary_from_cam(ctx, hr_dev, B, K, N, conf_bg_thres, GT_PROB, u_dev)
            crf.inference(u_dev, K + 1, g_compat, bi_compat, t, None, bg_dev)
        conf_dev = ctx.alloc(B * N, pooled=True)
        _lib.ir_label_combine(ctx, fg_dev, bg_dev, keys, N, conf_dev)
        conf = ctx.to_host(conf_dev, (B, H, W), np.uint8)
    finally:
        crf.close()
    return conf


def ir_label_ragged(ctx, items, mode, conf_fg_thres, conf_bg_thres):
    """A device batch exactly as the dataset yields it: items = [(img uint8 (H_b, W_b, 3), maps float32 (K_b, H_b, W_b),
    keys (K_b,))], every image with its own size and class count (cam_to_ir_label.py:25-58 runs them one by one) -> list of
    uint8 conf maps (H_b, W_b).  One wsc_crf_v for the batch; bit-identical to ir_label_batch on each (H, W, K) group."""
    g_sxy, g_compat, bi_sxy, bi_srgb, bi_compat, t = CRF_PARAMS
    B = len(items)
    sizes = [it[0].shape[:2] for it in items]
    Ns = [h * w for h, w in sizes]
    Ks = [int(np.asarray(it[1]).shape[0]) for it in items]
    rgb = [ctx.to_device(np.ascontiguousarray(it[0], dtype=np.uint8), pooled=True) for it in items]
    hr = [ctx.to_device(np.ascontiguousarray(it[1], dtype=np.float32), pooled=True) for it in items]
    crf = _lib.CrfV(ctx, rgb, sizes, g_sxy, bi_sxy, bi_srgb)
    try:
        u = [ctx.alloc((K + 1) * N * 4, pooled=True) for K, N in zip(Ks, Ns)]
        fg = [ctx.alloc(N * 4, pooled=True) for N in Ns]
        for b in range(B):
            _lib.label_unary_from_cam(ctx, hr[b], 1, Ks[b], Ns[b], conf_fg_thres, GT_PROB, u[b])
        crf.inference(u, [K + 1 for K in Ks], g_compat, bi_compat, t, None, fg)
        bg = [None] * B
        if mode == "voc12":
            bg = [ctx.alloc(N * 4, pooled=True) for N in Ns]
            for b in range(B):
                _lib.label_unary_from_cam(ctx, hr[b], 1, Ks[b], Ns[b], conf_bg_thres, GT_PROB, u[b])
            crf.inference(u, [K + 1 for K in Ks], g_compat, bi_compat, t, None, bg)
        offs = np.concatenate(([0], np.cumsum(Ns))).astype(np.int64)
        conf_dev = ctx.alloc(int(offs[-1]), pooled=True)
        for b, it in enumerate(items):
            k = np.asarray(it[2], dtype=np.int64)
            keys = np.pad(k + 1, (1, 0), mode="constant") if mode == "voc12" else np.concatenate((np.array([-1]), k))
            _lib.ir_label_combine(ctx, fg[b], bg[b], keys[None], Ns[b], conf_dev.ptr + int(offs[b]))
        flat = ctx.to_host(conf_dev, (int(offs[-1]),), np.uint8)
    finally:
        crf.close()
    return [flat[offs[b]:offs[b + 1]].reshape(sizes[b]).copy() for b in range(B)]


def ir_label_voc12(img, cam_dict, conf_fg_thres=0.30, conf_bg_thres=0.05, ctx=None):
    """cam_to_ir_label.py:42-58 for one VOC image: uint8 (H, W) label map, 255 = unreliable region."""
    ctx = ctx or imutils.default_context()
    img = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
    if len(cam_dict["keys"]) == 0:  # no foreground class: everything is background
        return np.zeros(img.shape[:2], np.uint8)
    return ir_label_batch(ctx, img[None], np.asarray(cam_dict["high_res"])[None], [cam_dict["keys"]], "voc12", conf_fg_thres,
                          conf_bg_thres)[0]


def _colour(args, conf):
    """cam_to_ir_label.py:79-88."""
    clr = np.zeros(conf.shape + (3,), dtype=np.uint8)
    off = 0
    for t in ("bg", "fg"):
        for i, c in enumerate(args.class_colours[t]):
            clr[conf == (i + off)] = np.asarray(c, dtype=np.uint8)
        off += len(args.class_colours[t])
    clr[conf == 255] = 255
    return clr


def _save(args, name, conf, img):
    from PIL import Image

    Image.fromarray(conf).save(os.path.join(args.ir_label_out_dir, name + ".png"))
    clr_dir = getattr(args, "ir_label_clr_out_dir", None)
    if clr_dir:
        clr = _colour(args, conf)
        Image.fromarray(clr).save(os.path.join(clr_dir, name + ".png"))
        over = np.uint8((1 - args.overlay_r) * np.float32(img) + args.overlay_r * np.float32(clr))  # :90-91
        Image.fromarray(over).save(os.path.join(clr_dir, name + "_overlay.png"))


def _work(process_id, infer_dataset, args):
    """cam_to_ir_label.py:18-95: reads <cam_out_dir>/<name>.npy of this worker's shard, writes the IR label PNGs."""
    databin = infer_dataset[process_id]
    ids = getattr(args, "cam_device_ids", None)
    ctx = _lib.Context(int(ids[process_id]) if ids is not None else process_id)
    voc = args.dataset == "voc12"
    dg = args.dataset in ("deepglobe", "deepglobe_balanced")
    if not (voc or dg or args.dataset in ("adp_morph", "adp_func")):
        raise KeyError("Dataset %s not yet implemented" % args.dataset)
    max_batch = int(getattr(args, "ir_label_batch_images", 16))
    pending = []  # [(name, img, maps, keys)] in dataset order: mixed sizes and class counts, one ragged CRF per device batch

    def flush():
        items = list(pending)
        del pending[:]
        conf = ir_label_ragged(ctx, [(it[1], it[2], it[3]) for it in items], "voc12" if voc else "fg", args.conf_fg_thres,
                               getattr(args, "conf_bg_thres", 0.05))
        for it, c in zip(items, conf):
            _save(args, it[0], c, it[1])

    try:
        for i in range(len(databin)):
            pack = databin[i]
            name = pack["name"]
            img = np.asarray(pack["img"], dtype=np.uint8)
            cam_dict = np.load(os.path.join(args.cam_out_dir, name + ".npy"), allow_pickle=True).item()
            if dg:
                img = _nearest_resize_cv2(img, (img.shape[0] // 4, img.shape[1] // 4))  # :63 (cv2 takes (w, h))
                maps = np.asarray(cam_dict.get("cam", np.empty(0)))
            else:
                maps = np.asarray(cam_dict.get("high_res", np.empty(0)))
            keys = np.asarray(cam_dict["keys"])
            if len(keys) == 0:  # nothing to refine: VOC -> all background; ADP / DeepGlobe -> keys = [-1] -> all 255
                _save(args, name, np.full(img.shape[:2], 0 if voc else 255, np.uint8), img)
                continue
            maps = maps.reshape((len(keys),) + img.shape[:2])
            pending.append((name, img, maps, keys))
            if len(pending) >= max_batch:
                flush()
        if pending:
            flush()
    finally:
        ctx.close()


def run(args):
    """cam_to_ir_label.py:98-117.  `args.dataset_obj` may hand in the image dataset (items {"name", "img" uint8 HWC});
    the default builds the reference's ImageDataset mirrors from args.train_list / args.dev_root."""
    from ..misc import torchutils

    for d in (args.ir_label_out_dir, getattr(args, "ir_label_clr_out_dir", None)):
        if d:
            os.makedirs(d, exist_ok=True)
    dataset = getattr(args, "dataset_obj", None)
    if dataset is None:
        if args.dataset == "voc12":
            from ..voc12 import dataloader

            dataset = dataloader.VOC12ImageDataset(args.train_list, dev_root=args.dev_root)
        elif args.dataset in ("adp_morph", "adp_func"):
            from ..adp import dataloader

            dataset = dataloader.ADPImageDataset(args.train_list, dev_root=args.dev_root, htt_type=args.dataset.split("_")[-1],
                                                 is_eval=args.split == "evaluation")
        elif args.dataset in ("deepglobe", "deepglobe_balanced"):
            from ..deepglobe import dataloader

            dataset = dataloader.DeepGlobeImageDataset(args.train_list, dev_root=args.dev_root,
                                                       is_balanced=args.dataset == "deepglobe_balanced")
        else:
            raise KeyError("Dataset %s not yet implemented" % args.dataset)
    n = max(1, int(getattr(args, "num_workers", 1)))
    shards = torchutils.split_dataset(dataset, n)
    if n == 1:
        _work(0, shards, args)
    else:
        import torch.multiprocessing as mp

        mp.spawn(_work, nprocs=n, args=(shards, args), join=True)
