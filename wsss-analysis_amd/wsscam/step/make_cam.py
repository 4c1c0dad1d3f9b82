"""make_cam step -- drop-in for 03b_irn/step/make_cam.py (run :95-124, _work :25-93).

Same `run(args)` contract: builds the CAM network named by args.cam_network, loads weights,
builds the MSF dataset, splits it round-robin over the GPUs and writes one
`<cam_out_dir>/<name>.npy` per image holding the pickled dict
{"keys": int64[K], "cam": float32[K,h4,w4], "high_res": float32[K,H0,W0]}
(`high_res` omitted for deepglobe, three empty arrays when no class is valid) that
eval_cam.py:48-58, cam_to_ir_label.py:27-33 and make_sem_seg_labels.py:61-66 consume.

What changes is underneath: instead of one image (2 samples) per launch, each process batches
`args.cam_batch_images` images (default 32 = 64 samples) through libwsscam -- conv stack, CAM head
+ flip-add, both bilinear resizes and the per-class max-normalisation all on the device -- and
only the final maps cross PCIe.  One process per GPU, disjoint shards, no collective
(make_cam.py:120-122).
"""
import importlib
import os

import numpy as np

from .. import _lib
from ..misc import torchutils


def _valid_cat(args, pack, score, model):
    """make_cam.py:49-54: GT image-level labels on 'train' splits, predicted labels otherwise."""
    if "train" in args.split:
        label = np.asarray(pack["label"])
    else:
        if score is None:
            raise ValueError("split %r needs predicted labels but %s has no classifier branch"
                             % (args.split, type(model).__name__))
        label = model.predict_labels(score)
        if getattr(args, "use_cls", None) is not None:
            label = label[args.use_cls]
    cat = np.nonzero(np.asarray(label))[0].astype(np.int64)
    if args.dataset in ("adp_morph", "adp_func"):  # make_cam.py:56-61: the synthesised bg channels always count
        n_bg = len(args.class_names["bg"])
        cat = np.concatenate((np.arange(n_bg, dtype=np.int64), cat + n_bg))
    return cat


def _npy_object_bytes(obj):
    """The bytes `np.save(path, obj)`-compatible container of a Python object: .npy header of a 0-d object array + a protocol-5
    pickle (in-band buffers)."""
    import io
    import pickle

    from numpy.lib import format as npy_format

    arr = np.empty((), dtype=object)
    arr[()] = obj
    bio = io.BytesIO()
    npy_format.write_array_header_1_0(bio, npy_format.header_data_from_array_1_0(arr))
    pickle.dump(arr, bio, protocol=5)
    return bio.getvalue()


_NPY_TEMPLATES = {}   # signature of a dict of arrays -> (metadata pieces, patched small arrays) of its .npy container
_NPY_TEMPLATES_MAX = 8192


def _npy_template(obj, sig):
    """Cuts the container of `obj` (a dict of C-contiguous arrays) into [piece 0 | array 0 | piece 1 | array 1 | ... | piece n]:
    the pieces (headers, pickle opcodes, frame lengths, dtype / shape tuples) depend on the dtypes, shapes and flags only, so
    the next dict of the same signature is written as the cached pieces around ITS arrays.  None when the cut cannot be
    verified (the caller then writes the plain pickle)."""
    import struct

    full = _npy_object_bytes(obj)
    pieces, at = [], 0
    for v in obj.values():
        raw = v.tobytes()
        n = len(raw)
        pos = at
        while True:
            pos = full.find(raw, pos) if n else -1
            if pos < 0:
                return None
            # an in-band buffer is preceded by its opcode and length: BYTEARRAY8 / BINBYTES8 (8-byte length), BINBYTES (4),
            # SHORT_BINBYTES (1) -- anything else is a chance match inside the metadata
            if (full[pos - 9:pos - 8] in (b"\x96", b"\x8e") and full[pos - 8:pos] == struct.pack("<Q", n)) or \
                    (full[pos - 5:pos - 4] == b"B" and full[pos - 4:pos] == struct.pack("<I", n)) or \
                    (full[pos - 2:pos - 1] == b"C" and full[pos - 1:pos] == struct.pack("<B", n & 0xff) and n < 256):
                break
            pos += 1
        pieces.append(full[at:pos])
        at = pos + n
    pieces.append(full[at:])
    # verification: the pieces around the arrays reproduce the container
    chk = b"".join(p + v.tobytes() for p, v in zip(pieces, obj.values())) + pieces[-1]
    return pieces if chk == full else None


def save_npy_object(path, obj):
    """`np.save(path, obj)` for a Python object (the reference stores a dict of arrays that way, make_cam.py:80-88), written so
    that writer THREADS scale: same .npy container (0-d object array, `np.load(path, allow_pickle=True).item()` gives the same
    dict, dtypes and shapes) with a protocol-5 pickle inside, whose in-band buffers are the arrays' own bytes.
    Round 4: pickle.dump straight to the file (np.save's protocol 3 copies every array into a bytes object with the interpreter
    lock held: 0.3 ms per image alone, 1.3 ms under contention).  Round 6: for a dict of non-empty C-contiguous arrays the
    metadata pieces of the container are cached per (keys, dtypes, shapes) signature and ONE C call (wsc_host_write_segments:
    open + writev + close, no interpreter lock) writes them around the arrays where they lie -- e.g. in the page-locked staging
    buffer of the D2H copy.  64 files per 32-image step held the lock for most of a step before (profiles/README.md, round 6).
    Anything else (other objects, empty arrays, a cut that does not verify) takes the pickle.dump path: same bytes."""
    if not str(path).endswith(".npy"):
        path = str(path) + ".npy"  # np.save appends the extension
    if isinstance(obj, dict) and obj and all(isinstance(v, np.ndarray) and v.flags["C_CONTIGUOUS"] and v.size > 0 and
                                             v.dtype.kind in "fiub" for v in obj.values()):
        sig = tuple((k, v.dtype.str, v.shape, bool(v.flags["WRITEABLE"])) for k, v in obj.items())
        pieces = _NPY_TEMPLATES.get(sig, False)
        if pieces is False:
            pieces = _npy_template(obj, sig)
            if len(_NPY_TEMPLATES) < _NPY_TEMPLATES_MAX:
                _NPY_TEMPLATES[sig] = pieces
        if pieces is not None:
            segs = []
            for pc, v in zip(pieces, obj.values()):
                segs.append(pc)
                segs.append(v)
            segs.append(pieces[-1])
            _lib.host_write_segments(path, segs)
            return
    with open(path, "wb") as fp:
        import pickle

        from numpy.lib import format as npy_format

        arr = np.empty((), dtype=object)
        arr[()] = obj
        npy_format.write_array_header_1_0(fp, npy_format.header_data_from_array_1_0(arr))
        pickle.dump(arr, fp, protocol=5)


_NPY_HEADERS = {}


def save_npy_array(path, arr):
    """`np.save(path, arr)` of a plain C-contiguous array as one C call (cached .npy header + the array's bytes)."""
    import io

    from numpy.lib import format as npy_format

    arr = np.ascontiguousarray(arr)
    if not str(path).endswith(".npy"):
        path = str(path) + ".npy"
    key = (arr.dtype.str, arr.shape)
    hdr = _NPY_HEADERS.get(key)
    if hdr is None:
        bio = io.BytesIO()
        npy_format.write_array_header_1_0(bio, npy_format.header_data_from_array_1_0(arr))
        hdr = bio.getvalue()
        if len(_NPY_HEADERS) < _NPY_TEMPLATES_MAX:
            _NPY_HEADERS[key] = hdr
    _lib.host_write_segments(path, [hdr, arr])


def _save(args, name, keys, strided, highres):
    path = os.path.join(args.cam_out_dir, name + ".npy")
    if len(keys) == 0:  # make_cam.py:86-88
        save_npy_object(path, {"keys": np.empty(0), "cam": np.empty(0), "high_res": np.empty(0)})
    elif args.dataset in ("deepglobe", "deepglobe_balanced"):  # make_cam.py:83-85
        save_npy_object(path, {"keys": keys, "cam": strided})
    else:  # make_cam.py:80-82
        save_npy_object(path, {"keys": keys, "cam": strided, "high_res": highres})


def _scales_of(v):
    """The per-scale entries of an MSF dataset field: a list for several scales, the array itself for one."""
    return list(v) if isinstance(v, (list, tuple)) else [v]


def process_batch(model, packs, args, save=True):
    """One device batch: list of dataset items -> list of result dicts (and .npy files)."""
    ctx = model.ctx
    B = len(packs)
    if B and "img" not in packs[0]:
        # items of a dataset built with device_transform=True (decoded uint8 images) on the SERIAL path (cam_pipeline off):
        # the same transform on the host -- msf_pack is what wsc_msf_input_u8 reproduces bit for bit (tests/test_gpu_input.py)
        from ..voc12.dataloader import TorchvisionNormalize, msf_pack

        norm = TorchvisionNormalize(getattr(args, "norm_mode", "int"))
        outsize = getattr(args, "outsize", (321, 321))
        conv = []
        for p in packs:
            ms = [msf_pack(np.asarray(si), outsize, norm) for si in _scales_of(p["img_u8"])]
            conv.append(dict(p, img=ms[0] if len(ms) == 1 else ms))
        packs = conv
    # multi-scale items carry a list of (2,3,S,S) pairs (one per args.cam_scales entry, all resized to the same S): the scales
    # of an image run as consecutive "images" of one device batch and their CAMs are summed before the tail
    n_sc = len(_scales_of(packs[0]["img"]))
    shapes_in = [tuple(np.asarray(_scales_of(p["img"])[0]).shape) for p in packs]
    if len(set(shapes_in)) > 1:
        # outsize = None (the reference's resnet50 configuration, func_sample.py:143-148): every image keeps its own size, so
        # the items of a batch are bucketed by network input size -- one device batch per size, results in input order
        if n_sc > 1:
            raise ValueError("make_cam: several cam_scales need one network input size per run (args.outsize)")
        out = [None] * B
        buckets = {}
        for i, sh in enumerate(shapes_in):
            buckets.setdefault(sh, []).append(i)
        for idx in buckets.values():
            for i, o in zip(idx, process_batch(model, [packs[i] for i in idx], args, save=save)):
                out[i] = o
        return out
    x = np.stack([np.stack([np.asarray(v, dtype=np.float32) for v in _scales_of(p["img"])]) for p in packs])  # (B,n_sc,2,3,H,W)
    if x.ndim != 6 or x.shape[2:4] != (2, 3):
        raise ValueError("make_cam: network inputs must be (B, 2, 3, H, W); got %s" % (x.shape,))
    S, SW = int(x.shape[-2]), int(x.shape[-1])  # square for args.outsize = (321, 321) / (224, 224); the image's size for None
    C = model.num_classes
    h, w = model.cam_size_hw(S, SW)
    has_cls = model.arch != _lib.ARCH_RESNET50_CAM
    x_dev = ctx.to_device(x)
    cam_dev = ctx.alloc(B * n_sc * C * h * w * 4)
    score_dev = ctx.alloc(B * n_sc * C * 4) if has_cls else None
    model.forward_batch_device(x_dev, B * n_sc, S, cam_dev, score_dev, SW=SW)
    # make_cam.py:52: `label = labels[0][args.use_cls]` -- the prediction of the FIRST scale decides the classes
    score = ctx.to_host(score_dev, (B, n_sc, C), np.float32)[:, 0] if has_cls else None
    keys = [_valid_cat(args, p, None if score is None else score[b], model) for b, p in enumerate(packs)]
    sizes = [tuple(int(v) for v in p["size"]) for p in packs]
    if args.dataset in ("adp_morph", "adp_func"):
        # vgg16_cam.py:51-58 / common_cam.py:31-92: background (and 'other') channels are synthesised from the ORIGINAL
        # (un-flipped) image of every scale and joined with the use_cls CAM channels before the tail -- on the device
        # (wsc_hsn_background + wsc_cam_adp_modify): the CAM stack never visits the host
        origs = [np.asarray(o)[0] for p in packs for o in _scales_of(p["orig_img"])]
        cam_dev, C = model.adp_modify_device(ctx, cam_dev, B, n_sc, h, w, origs)
    elif n_sc > 1:
        sum_dev = ctx.alloc(B * C * h * w * 4)
        _lib.cam_sum_scales(ctx, cam_dev, B, n_sc, C * h * w, sum_dev)
        cam_dev = sum_dev
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, B, C, h, w, sizes, keys)
    s_tot = sum(k * a * b for (k, a, b, _, _) in shapes)
    h_tot = sum(k * a * b for (k, _, _, a, b) in shapes)
    strided = ctx.to_host(s_dev, (max(s_tot, 1),), np.float32)
    highres = ctx.to_host(h_dev, (max(h_tot, 1),), np.float32)
    out = []
    for b, p in enumerate(packs):
        K, h4, w4, H0, W0 = shapes[b]
        sc = strided[s_off[b]:s_off[b] + K * h4 * w4].reshape(K, h4, w4).copy()
        hc = highres[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0).copy()
        if save:
            _save(args, p["name"], keys[b], sc, hc)
        out.append({"keys": keys[b], "cam": sc, "high_res": hc})
    return out


def _work(process_id, model, dataset, args):
    """Per-GPU worker (make_cam.py:25-93): shard `process_id` of the dataset on device `process_id`."""
    databin = dataset[process_id]
    # device of this worker: process_id as in the reference (cuda.device(process_id), make_cam.py:31); `cam_device_ids`
    # remaps it (e.g. two workers on one GPU)
    ids = getattr(args, "cam_device_ids", None)
    device = int(ids[process_id]) if ids is not None else process_id
    model.cuda(device)
    bs = int(getattr(args, "cam_batch_images", 32))
    n = len(databin)
    if n == 0:
        return
    native = getattr(args, "outsize", (321, 321)) is None  # every image at its own size: batches bucketed by size, no pinned lanes
    if native or not getattr(args, "cam_pipeline", True):
        for i0 in range(0, n, bs):
            packs = [databin[i] for i in range(i0, min(i0 + bs, n))]
            process_batch(model, packs, args, save=True)
        model.ctx.sync()
        return
    from .pipeline import CamPipeline

    first = databin[0]
    norm = None
    n_sc = len(_scales_of(first["img"] if "img" in first else first["img_u8"]))
    if "img" in first:
        f0 = np.asarray(_scales_of(first["img"])[0])
        S = int(f0.shape[-1])
        if f0.shape != (2, 3, S, S):
            raise ValueError("make_cam: network inputs must be (2, 3, S, S) with one square size per run; got %s" % (f0.shape,))
    else:  # dataset built with device_transform=True: items carry the decoded image, the transform runs on the GPU
        S = int(args.outsize[0])
        base = getattr(databin, "dataset", databin)
        norm = getattr(base, "norm", None)
        if norm is None:
            from ..voc12.dataloader import TorchvisionNormalize

            norm = TorchvisionNormalize(args.norm_mode)
    has_cls = model.arch != _lib.ARCH_RESNET50_CAM
    needs_score = "train" not in args.split
    if needs_score and not has_cls:
        raise ValueError("split %r needs predicted labels but %s has no classifier branch" % (args.split, type(model).__name__))
    # loader / writer pools: sized from the host cores per worker unless the caller fixes them (pipeline.host_thread_budget)
    lt, wt = getattr(args, "cam_loader_threads", None), getattr(args, "cam_writer_threads", None)
    pipe = CamPipeline(model, device, bs, S, keys_fn=lambda pack, score: _valid_cat(args, pack, score, model),
                       save_fn=lambda name, keys, sc, hc: _save(args, name, keys, sc, hc), needs_score=needs_score,
                       n_lanes=int(getattr(args, "cam_pipeline_lanes", 3)), n_loaders=None if lt is None else int(lt),
                       n_writers=None if wt is None else int(wt), norm=norm, n_scales=n_sc,
                       world=int(getattr(args, "n_gpus", 0)) or len(dataset),
                       chain_stacks=bool(getattr(args, "cam_pipeline_chain", True)))
    try:
        pipe.run(databin)
    finally:
        pipe.close()


def _device_count():
    import torch

    return torch.cuda.device_count()


def _device_transform_default(args):
    """args.cam_device_transform when given; otherwise ON for multi-GPU runs: N workers share the host, and the decoded uint8
    image is 8x fewer PCIe bytes and no float64 host resize per image (wsc_msf_input_u8 is bit-identical to the host transform,
    tests/test_gpu_input.py).  A single worker keeps the host transform (the reference's data path)."""
    v = getattr(args, "cam_device_transform", None)
    if v is not None:
        return bool(v)
    # only where the overlapped pipeline (the consumer of "img_u8" items on the device) will run: voc12 with a fixed network
    # size; the worker count is derived exactly as run() derives it (args.n_gpus, else every visible device)
    if not getattr(args, "cam_pipeline", True) or getattr(args, "outsize", (321, 321)) is None:
        return False
    n = int(getattr(args, "n_gpus", 0) or 0) or _device_count()
    return n > 1


def build_dataset(args):
    if getattr(args, "dataset_obj", None) is not None:
        return args.dataset_obj
    if args.dataset == "voc12":
        from ..voc12 import dataloader

        return dataloader.VOC12ClassificationDatasetMSF(args.val_list, norm_mode=args.norm_mode,
                                                        outsize=args.outsize, dev_root=args.dev_root,
                                                        scales=args.cam_scales,
                                                        cls_labels_path=getattr(args, "cls_labels_path", None),
                                                        device_transform=_device_transform_default(args))
    if args.dataset in ("adp_morph", "adp_func"):
        from ..adp import dataloader

        return dataloader.ADPClassificationDatasetMSF(args.val_list, norm_mode=args.norm_mode, outsize=args.outsize,
                                                      dev_root=args.dev_root, htt_type=args.dataset.split("_")[-1],
                                                      is_eval=args.split == "evaluation", scales=args.cam_scales,
                                                      cls_labels_path=getattr(args, "cls_labels_path", None))
    if args.dataset in ("deepglobe", "deepglobe_balanced"):
        from ..deepglobe import dataloader

        return dataloader.DeepGlobeClassificationDatasetMSF(args.val_list, norm_mode=args.norm_mode,
                                                            outsize=args.outsize, dev_root=args.dev_root,
                                                            is_balanced=args.dataset == "deepglobe_balanced",
                                                            scales=args.cam_scales,
                                                            cls_labels_path=getattr(args, "cls_labels_path", None))
    raise KeyError("Dataset %s not yet implemented" % args.dataset)


def run(args):
    """03b_irn/step/make_cam.py:95-124."""
    mod = args.cam_network
    if not mod.startswith("wsscam."):
        mod = "wsscam." + mod  # the reference passes 'net.resnet50_cam'
    model = getattr(importlib.import_module(mod), "CAM")(args.model_dir, args.dataset, args.tag,
                                                         args.num_classes, args.use_cls)
    if getattr(args, "cam_precision", None) is not None:  # optional: _lib.PREC_F16X3 (default, fp32-class) / F16 / BF16 / BF16X3
        model.precision = args.cam_precision
    if getattr(args, "state_dict", None) is not None:
        model.load_state_dict(args.state_dict, strict=True)  # weights handed over in memory (tests, dry runs)
    elif getattr(args, "model_id", None) == "resnet50":      # make_cam.py:98-99: only resnet50 has a .pth
        import torch

        model.load_state_dict(torch.load(args.cam_weights_name + ".pth", map_location="cpu"), strict=True)
    # the vgg16 / m7 / x1.7 wrappers transplanted their Keras weights from args.model_dir in the constructor
    # (common_cnn._load_pretrained); a wrapper that is still empty fails loudly at the first forward pass
    model.eval()

    n_gpus = int(getattr(args, "n_gpus", 0)) or _device_count()
    if n_gpus < 1:
        raise _lib.WscError(_lib.WSC_ERR_NO_DEVICE, "make_cam needs at least one gfx950 device")
    os.makedirs(args.cam_out_dir, exist_ok=True)
    dataset = torchutils.split_dataset(build_dataset(args), n_gpus)

    if n_gpus == 1:
        _work(0, model, dataset, args)
    else:
        # one OS process per GPU, as multiprocessing.spawn(_work, nprocs=n_gpus) does (make_cam.py:122);
        # each child creates its own HIP context after the fork-less spawn
        import torch.multiprocessing as mp

        mp.spawn(_work, nprocs=n_gpus, args=(model, dataset, args), join=True)
