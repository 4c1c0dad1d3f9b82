"""GPU: edge cases of the hot path -- tiny / ragged / empty inputs, native VOC sizes, the end-to-end
make_cam.run drop-in with its on-disk format."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import cnn_ref
from tests import helpers
from wsscam import _lib
from wsscam.hsn import utilities as hsn_utilities
from wsscam.step import make_cam

pytestmark = pytest.mark.gpu


def test_tiny_network_input_single_image():
    """B = 1, S = 33: every layer has fewer output pixels than one 128-row tile by the end (3x3 maps)."""
    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    from wsscam.net import resnet50_cam

    m = resnet50_cam.CAM(None, "voc12", "", 20, None, precision=_lib.PREC_BF16X3)
    m.load_state_dict(sd)
    m.cuda(0)
    rng = np.random.default_rng(1)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 40, 50), (33, 33))
    cam = m.forward(x)
    with torch.no_grad():
        ref = cnn_ref.resnet50_cam_forward(torch.from_numpy(x), sd).numpy()
    assert cam.shape == ref.shape == (20, 3, 3)
    assert np.abs(cam - ref).max() <= 2e-4 * ref.max()


def test_make_cam_run_end_to_end(tmp_path):
    """step.make_cam.run(args) with the reference's Namespace fields: one .npy per image in the
    reference's three layouts (make_cam.py:80-88), values vs the oracle's make_cam_image."""
    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    rng = np.random.default_rng(2)
    sizes = [(60, 80), (97, 64), (64, 64), (33, 47), (80, 60)]
    labels = [np.zeros(20, np.float32) for _ in sizes]
    labels[0][[1, 4]] = 1
    labels[1][[7]] = 1
    # labels[2] stays empty: the reference writes three empty arrays
    labels[3][[0, 19]] = 1
    labels[4][[12]] = 1
    data = [{"name": "2007_%06d" % i, "img": cnn_ref.msf_pack(cnn_ref.synth_image(rng, *sz), (65, 65)), "size": sz,
             "label": lb} for i, (sz, lb) in enumerate(zip(sizes, labels))]
    args = types.SimpleNamespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                                 use_cls=None, model_id="resnet50", cam_weights_name=None, state_dict=sd,
                                 dataset_obj=data, split="train_aug", cam_out_dir=str(tmp_path), n_gpus=1,
                                 cam_batch_images=2,  # 5 images in batches of 2: a ragged last batch
                                 cam_precision=_lib.PREC_BF16X3)  # 5x5 CAMs at S=65: check the plumbing tightly
    make_cam.run(args)
    files = sorted(os.listdir(tmp_path))
    assert files == [d["name"] + ".npy" for d in data]
    for d in data:
        rec = np.load(os.path.join(tmp_path, d["name"] + ".npy"), allow_pickle=True).item()
        ref = cnn_ref.make_cam_image(torch.from_numpy(d["img"]), sd, d["size"], torch.from_numpy(d["label"]))
        assert list(rec) == ["keys", "cam", "high_res"]
        if d["label"].sum() == 0:
            assert all(rec[k].shape == (0,) for k in rec)
            continue
        assert rec["keys"].dtype == np.int64 and np.array_equal(rec["keys"], ref["keys"])
        assert rec["cam"].shape == ref["cam"].shape and rec["high_res"].shape == ref["high_res"].shape
        assert np.abs(rec["high_res"] - ref["high_res"]).max() <= 2e-4
        assert np.abs(rec["cam"] - ref["cam"]).max() <= 2e-4


@pytest.mark.parametrize("mode", ["batch", "pipeline", "pipeline_u8"])
def test_make_cam_multi_scale(tmp_path, mode):
    """args.cam_scales with several entries (make_cam.py:62-69 sums the per-scale interpolated maps; the MSF dataset rescales
    with PIL bicubic and resizes every scale to the same network input, voc12/dataloader.py:231-240): dataset items with a
    LIST of pairs through the serial path, the overlapped pipeline and the device-side input transform, against the oracle
    in the reference's order (interpolate each scale, then sum).  Summing before the (linear) interpolation differs by
    fp32 rounding only: the bound is the single-scale one."""
    from wsscam.misc import imutils
    from wsscam.voc12 import dataloader as voc_dl

    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    rng = np.random.default_rng(5)
    sizes = [(60, 80), (97, 64), (64, 72), (50, 50), (33, 47)]
    scales = (1.0, 0.5, 1.5)
    labels = [np.zeros(20, np.float32) for _ in sizes]
    for i, cls in enumerate([[1, 4], [7], [], [0, 19], [12]]):
        labels[i][cls] = 1
    norm = voc_dl.TorchvisionNormalize("float")
    raws = [cnn_ref.synth_image(rng, *sz) for sz in sizes]
    data, data_u8 = [], []
    for i, (raw, sz, lb) in enumerate(zip(raws, sizes, labels)):
        s_imgs = imutils.scale_images(raw, scales)
        assert s_imgs[1].shape[:2] == (int(np.round(sz[0] * 0.5)), int(np.round(sz[1] * 0.5))) and s_imgs[0] is raw
        data.append({"name": "2008_%06d" % i, "img": [voc_dl.msf_pack(si, (65, 65), norm) for si in s_imgs], "size": sz, "label": lb})
        data_u8.append({"name": "2008_%06d" % i, "img_u8": s_imgs, "size": sz, "label": lb})
    args = types.SimpleNamespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                                 use_cls=None, model_id="resnet50", cam_weights_name=None, state_dict=sd,
                                 dataset_obj=data_u8 if mode == "pipeline_u8" else data, split="train_aug",
                                 cam_out_dir=str(tmp_path), n_gpus=1, cam_batch_images=2, cam_precision=_lib.PREC_BF16X3,
                                 cam_pipeline=mode != "batch", outsize=(65, 65), norm_mode="float", cam_scales=scales)
    make_cam.run(args)
    contrib = 0.0
    for d in data:
        rec = np.load(os.path.join(tmp_path, d["name"] + ".npy"), allow_pickle=True).item()
        if d["label"].sum() == 0:
            assert all(rec[k].shape == (0,) for k in rec)
            continue
        valid = torch.nonzero(torch.from_numpy(d["label"]))[:, 0]
        with torch.no_grad():
            cams = [cnn_ref.resnet50_cam_forward(torch.from_numpy(x), sd) for x in d["img"]]
            rs, rh = cnn_ref.make_cam_tail(cams, d["size"], valid)
            one, _ = cnn_ref.make_cam_tail(cams[0], d["size"], valid)
        assert np.array_equal(rec["keys"], valid.numpy())
        assert np.abs(rec["cam"] - rs.numpy()).max() <= 2e-4 and np.abs(rec["high_res"] - rh.numpy()).max() <= 2e-4
        contrib = max(contrib, float(np.abs(rs.numpy() - one.numpy()).max()))
    assert contrib > 1e-2  # the other scales really contribute (a random-init class map can be all zero: checked over the set)


def test_cam_tail_large_native_size(ctx):
    """ADP evaluation size 1088x1088 (eval_cam.py:28): 1.18 M pixels per class map."""
    rng = np.random.default_rng(3)
    cam = np.maximum(rng.normal(0.4, 1.0, (1, 5, 40, 40)), 0).astype(np.float32)
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, ctx.to_device(cam), 1, 5, 40, 40, [(1088, 1088)],
                                                              [[1, 3]])
    K, h4, w4, H0, W0 = shapes[0]
    hi = ctx.to_host(h_dev, (K, H0, W0), np.float32)
    rs, rh = cnn_ref.make_cam_tail(torch.from_numpy(cam[0]), (1088, 1088), torch.tensor([1, 3]))
    assert np.abs(hi - rh.numpy()).max() <= 2e-6


@pytest.mark.parametrize("size", [(321, 321), (97, 130), (16, 33)])
def test_cam_unary_fused_equals_two_step_and_oracle(ctx, size):
    """wsc_cam_unary = wsc_cam_postprocess (all classes) + wsc_unary_from_maps without the maps in HBM: bit-identical
    to the two-step path, and both within 2e-5 of the torch/numpy restatement (make_cam.py:64-76, eval_cam.py:49-51,
    unary_from_softmax); one all-zero class map (0 / 1e-5 = 0 -> the 1e-5 clip) included."""
    B, C, h, w = 3, 20, 21, 21
    H0, W0 = size
    rng = np.random.default_rng(H0 * 1000 + W0)
    cam = np.maximum(rng.normal(0.3, 1.0, (B, C, h, w)), 0).astype(np.float32)
    cam[1, 7] = 0.0
    cam_dev = ctx.to_device(cam)
    n = H0 * W0
    _, h_dev, _, _, _ = _lib.cam_postprocess(ctx, cam_dev, B, C, h, w, [size] * B, [list(range(C))] * B)
    u2_dev = ctx.alloc(B * (C + 1) * n * 4)
    _lib.unary_from_maps(ctx, h_dev, B, C, n, 0.15, u2_dev)
    u1_dev = ctx.alloc(B * (C + 1) * n * 4)
    _lib.cam_unary(ctx, cam_dev, B, C, h, w, H0, W0, 0.15, u1_dev)
    u1 = ctx.to_host(u1_dev, (B, C + 1, n), np.float32)
    u2 = ctx.to_host(u2_dev, (B, C + 1, n), np.float32)
    assert np.array_equal(u1, u2)
    for b in range(B):
        _, rh = cnn_ref.make_cam_tail(torch.from_numpy(cam[b]), size, torch.arange(C))
        v = np.concatenate([np.full((1, n), 0.15, np.float32), rh.numpy().reshape(C, n)], axis=0)
        ref = -np.log(np.clip(v / v.sum(0, keepdims=True), 1e-5, 1.0))
        assert np.abs(u1[b] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", [(20, 21, 321, 321), (20, 21, 97, 130), (20, 21, 16, 33), (7, 5, 1, 1), (3, 4, 1, 63),
                                  (30, 21, 70, 65), (32, 9, 33, 64)])
def test_cam_unary_pixel_major_rows_equal_class_major(ctx, case):
    """wsc_cam_unary_pm writes the rows of a wave through an LDS slab as contiguous kilobytes (or, when the slabs do not fit
    beside the class maps -- the 30 x 21 x 21 case -- with strided stores): same bits as the class-major kernel, padding
    columns zero, nothing written past the last pixel (sizes that are not multiples of a wave of 64 pixels)."""
    C, h, H0, W0 = case
    B = 2
    rng = np.random.default_rng(C * 100 + H0 + W0)
    cam = np.maximum(rng.normal(0.3, 1.0, (B, C, h, h)), 0).astype(np.float32)
    cam_dev = ctx.to_device(cam)
    n, M = H0 * W0, C + 1
    Mp = (M + 3) // 4 * 4
    u_cm = ctx.alloc(B * M * n * 4)
    guard = 64 * Mp
    sentinel = np.full(B * n * Mp + guard, -7.5, np.float32)
    u_pm = ctx.to_device(sentinel)
    _lib.cam_unary(ctx, cam_dev, B, C, h, h, H0, W0, 0.15, u_cm)
    _lib.cam_unary(ctx, cam_dev, B, C, h, h, H0, W0, 0.15, u_pm, pixel_major=True)
    U = ctx.to_host(u_cm, (B, M, n), np.float32)
    raw = ctx.to_host(u_pm, (B * n * Mp + guard,), np.float32)
    Upm = raw[: B * n * Mp].reshape(B, n, Mp)
    assert np.array_equal(np.transpose(Upm[:, :, :M], (0, 2, 1)), U) and np.all(Upm[:, :, M:] == 0)
    assert np.all(raw[B * n * Mp:] == -7.5)


def _gpu_crf(ctx, rgb, U, cfg):
    H, W, _ = rgb.shape
    M = U.shape[0]
    crf = _lib.Crf(ctx, ctx.to_device(rgb), 1, H, W, cfg[0], cfg[2], cfg[3])
    q_dev, a_dev = ctx.alloc(M * H * W * 4), ctx.alloc(H * W * 4)
    crf.inference(ctx.to_device(U), M, cfg[1], cfg[4], int(cfg[5]), q_dev, a_dev)
    q = ctx.to_host(q_dev, (M, H * W), np.float32)
    a = ctx.to_host(a_dev, (H * W,), np.int32)
    vg, vb = crf.lattice_sizes()
    crf.close()
    return q, a, (int(vg[0]), int(vb[0]))


@pytest.mark.parametrize("shape", [(1, 1), (1, 37), (29, 1), (3, 5)])
def test_crf_degenerate_sizes(ctx, shape):
    rng = np.random.default_rng(4)
    H, W = shape
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    p = rng.random((3, H * W)).astype(np.float32) + 0.05
    U = np.ascontiguousarray(-np.log(p / p.sum(0, keepdims=True)))
    cfg = (1.5, 3, 40, 13, 10, 3)
    q, a, v = _gpu_crf(ctx, rgb, U, cfg)
    qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
    assert v == (ls[0], ls[1])
    assert np.abs(q - qr).max() <= 1e-3 and np.array_equal(a, ar)


def test_crf_noise_image_hash_table_fallback(ctx):
    """Uniform-noise RGB: almost every pixel owns its six bilateral vertices, far more than the right-sized
    hash table (N (d+1) / 8 slots) holds -> the build flags the overflow and repeats with the worst-case
    table; a batch mixing a noise image with a smooth one takes the same path."""
    rng = np.random.default_rng(41)
    H, W, M = 64, 72, 3
    noise = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    smooth, U, _ = helpers.synth_crf_case(rng, H, W, M)
    cfg = (3, 3, 50, 5, 10, 3)
    q, a, v = _gpu_crf(ctx, noise, U, cfg)
    qr, ar, ls = helpers.crf_oracle(noise, U, cfg)
    assert v == (ls[0], ls[1]) and ls[1] > H * W  # more vertices than pixels
    assert np.abs(q - qr).max() <= 1e-3 and (a == ar).mean() >= 0.995
    from tests.test_gpu_crf import _gpu_crf as gpu_crf_batch

    qb, ab, vg, vb = gpu_crf_batch(ctx, [smooth, noise], [U, U], cfg)
    qs, as_, ls_s = helpers.crf_oracle(smooth, U, cfg)
    assert (vb[0], vb[1]) == (ls_s[1], ls[1])
    assert np.abs(qb[0] - qs).max() <= 1e-3 and np.abs(qb[1] - qr).max() <= 1e-3


def test_crf_native_voc_size_label_unaries(ctx):
    """cam_to_ir_label's regime: native 375x500, M = K+1 = 3, label unaries, irn CRF parameters."""
    from wsscam.misc import imutils

    rng = np.random.default_rng(5)
    H, W = 375, 500
    rgb, _, p = helpers.synth_crf_case(rng, H, W, 3)
    U = np.ascontiguousarray(imutils.unary_from_labels(p.argmax(0), 3, 0.7, zero_unsure=False))
    cfg = (3, 3, 50, 5, 10, 10)
    q, a, v = _gpu_crf(ctx, rgb, U, cfg)
    qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
    assert v == (ls[0], ls[1])
    assert np.abs(q - qr).max() <= 1e-3 and (a == ar).mean() >= 0.995


@pytest.mark.parametrize("case", [(321, 321, 29, (1.5, 3, 40, 13, 10, 10)),     # HSN ADP-morph regime: M = 29
                                  (1088, 1088, 5, (1.5, 3, 40, 13, 10, 10)),    # BASELINE config 5 stress: N = 1 183 744
                                  (41, 41, 21, (3 / 12, 3, 80 / 12, 13, 10, 5))])  # SEC/DSRG training-time CRF
def test_crf_config5_sizes(ctx, case):
    H, W, M, cfg = case
    rng = np.random.default_rng(50 + M)
    rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
    q, a, v = _gpu_crf(ctx, rgb, U, cfg)
    qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
    assert v == (ls[0], ls[1])
    assert np.abs(q - qr).max() <= 1e-3, np.abs(q - qr).max()
    assert (a == ar).mean() >= 0.995
    assert np.abs(q.sum(0) - 1).max() <= 1e-5


def test_dcrf_process_image_without_mass(ctx):
    """An image whose probabilities are all zero has no pass classes: its CRF output stays zero -> label 0."""
    rng = np.random.default_rng(6)
    H, W = 21, 25
    rgb, _, p = helpers.synth_crf_case(rng, H, W, 2)
    probs = np.zeros((2, 4, H, W))
    probs[0, [1, 3]] = p
    imgs = np.stack([rgb, rgb])
    out = hsn_utilities.dcrf_process(probs, imgs, [1.5, 3, 40, 13, 10, 5], ctx=ctx)
    assert np.array_equal(out[1], np.zeros((H, W), np.int64))
    assert set(np.unique(out[0])) <= {1, 3}


def test_argument_errors(ctx):
    with pytest.raises(_lib.WscError) as ei:
        _lib.Crf(ctx, ctx.to_device(np.zeros((4, 4, 3), np.uint8)), 1, 4, 4, -1.0, 40, 13)
    assert ei.value.status == _lib.WSC_ERR_INVALID
    crf = _lib.Crf(ctx, ctx.to_device(np.zeros((4, 4, 3), np.uint8)), 1, 4, 4, 1.5, 40, 13)
    with pytest.raises(_lib.WscError) as ei:
        crf.inference(ctx.alloc(33 * 16 * 4), 33, 3, 10, 1, None, None)  # M > 32
    assert ei.value.status == _lib.WSC_ERR_INVALID
    crf.close()
    with pytest.raises(_lib.WscError):
        _lib.cam_postprocess(ctx, ctx.alloc(4 * 21 * 21 * 4), 1, 4, 21, 21, [(10, 10)], [[7]])  # key out of range
    with pytest.raises(_lib.WscError) as ei:  # more classes than the fused kernel keeps in registers
        _lib.cam_unary(ctx, ctx.alloc(33 * 8 * 8 * 4), 1, 33, 8, 8, 16, 16, 0.15, ctx.alloc(34 * 256 * 4))
    assert ei.value.status == _lib.WSC_ERR_INVALID
    with pytest.raises(_lib.WscError):  # background value must be positive (it is a probability mass)
        _lib.cam_unary(ctx, ctx.alloc(4 * 8 * 8 * 4), 1, 4, 8, 8, 16, 16, 0.0, ctx.alloc(5 * 256 * 4))


def _adp_like_image(rng, H, W):
    """near-white background with pink/purple blobs: the expit(4 (mean - 240)) term is exercised."""
    img = np.full((H, W, 3), 244.0) + rng.normal(0, 3, (H, W, 3))
    yy, xx = np.mgrid[:H, :W]
    for _ in range(4):
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(5, 14)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img[m] = rng.uniform([150, 60, 130], [220, 130, 200])
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("dataset", ["adp_func", "adp_morph", "deepglobe"])
def test_make_cam_run_adp_and_deepglobe(tmp_path, dataset):
    """make_cam.run on the VGG16 networks of the other datasets: ADP joins background / other channels synthesised on
    the device (round 6: through the overlapped pipeline, like VOC) with the use_cls CAM channels before the tail and
    always keeps them (make_cam.py:44-61, vgg16_cam.py:51-58); DeepGlobe writes no high_res (make_cam.py:83-85)."""
    from wsscam.adp import dataloader as adp_dl

    adp = dataset.startswith("adp")
    C = 31 if adp else 6
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, not adp, seed=3)  # ADP models have no BN
    rng = np.random.default_rng(6)
    S = 65
    sizes = [(72, 72), (65, 65), (90, 70)]
    norm = adp_dl.TorchvisionNormalize("int") if adp else None
    data = []
    for i, sz in enumerate(sizes):
        img = _adp_like_image(rng, *sz) if adp else cnn_ref.synth_image(rng, *sz)
        if adp:
            x, orig = adp_dl.msf_item(img, (S, S), norm)
            lab = np.zeros(28 if dataset == "adp_morph" else 3, np.float32)
            lab[[i % len(lab), (3 * i + 1) % len(lab)]] = 1
            data.append({"name": "%03d.png_crop_%d" % (i, i), "img": x, "orig_img": orig, "size": sz, "label": lab})
        else:
            lab = np.zeros(6, np.float32)
            lab[[i, 5 - i]] = 1
            data.append({"name": "%06d" % (1000 + i), "img": cnn_ref.msf_pack(img, (S, S)), "size": sz, "label": lab})
    if dataset == "adp_morph":
        use_cls, bg_names = list(range(28)), ["background"]
    elif dataset == "adp_func":
        use_cls, bg_names = [28, 29, 30], ["background", "other"]
    else:
        use_cls, bg_names = list(range(6)), []
    args = types.SimpleNamespace(cam_network="net.vgg16_cam", model_dir=None, dataset=dataset, tag="", num_classes=C,
                                 use_cls=use_cls, model_id="vgg16", cam_weights_name=None, state_dict=sd,
                                 dataset_obj=data, split="train", cam_out_dir=str(tmp_path), n_gpus=1,
                                 cam_batch_images=2, cam_precision=_lib.PREC_BF16X3,
                                 class_names={"bg": bg_names, "fg": ["c%d" % i for i in range(len(use_cls))]})
    make_cam.run(args)
    for d in data:
        rec = np.load(os.path.join(tmp_path, d["name"] + ".npy"), allow_pickle=True).item()
        with torch.no_grad():
            cam, _ = cnn_ref.vgg16_cam_forward(torch.from_numpy(d["img"]), sd, C)
        cam = cam.numpy()
        if adp:  # common_cam.py:31-92 on the oracle CAM (oracle/cnn_ref.py::adp_modify)
            cam = cnn_ref.adp_modify(cam, d["orig_img"], dataset, use_cls)
        n_bg = len(bg_names)
        keys = np.concatenate((np.arange(n_bg), np.nonzero(d["label"])[0] + n_bg)).astype(np.int64)
        s, h = cnn_ref.make_cam_tail(torch.from_numpy(cam), d["size"], torch.from_numpy(keys))
        assert list(rec) == (["keys", "cam"] if dataset == "deepglobe" else ["keys", "cam", "high_res"])
        assert np.array_equal(rec["keys"], keys)
        assert rec["cam"].shape == tuple(s.shape)
        assert np.abs(rec["cam"] - s.numpy()).max() <= 5e-4, np.abs(rec["cam"] - s.numpy()).max()
        if dataset != "deepglobe":
            assert np.abs(rec["high_res"] - h.numpy()).max() <= 5e-4


@pytest.mark.parametrize("x17", [False, True])
def test_adp_channel_synthesis_on_device(ctx, x17):
    """common_cam.py:31-92 on the device (wsc_hsn_background + wsc_cam_adp_modify, the path of make_cam and of CAM.forward)
    against the oracle's numpy / scipy statement: morph and func stacks, the X1.7 class filter composed into the channel
    lists, two scales summed in scale order, original images of different sizes (one of them at the CAM size: no resize)."""
    from wsscam.net import vgg16_cam

    rng = np.random.default_rng(3)
    C, h, w, B, n_sc = (51 if x17 else 31), 12, 9, 3, 2
    cam = rng.random((B, n_sc, C, h, w)).astype(np.float32)
    shapes = [(40, 33), (40, 33), (12, 9), (12, 9), (25, 50), (31, 17)]
    origs = [_adp_like_image(rng, *sh) for sh in shapes]
    cam_dev = ctx.to_device(cam)
    for dataset, use_cls in (("adp_morph", list(range(28))), ("adp_func", [28, 29, 30])):
        m = vgg16_cam.CAM(None, dataset, "X1.7" if x17 else "", C, use_cls)
        out_dev, Cout = m.adp_modify_device(ctx, cam_dev, B, n_sc, h, w, origs)
        got = ctx.to_host(out_dev, (B, Cout, h, w), np.float32)
        assert Cout == m.adp_out_channels() == len(use_cls) + (1 if dataset == "adp_morph" else 2)
        for b in range(B):
            ref = sum(cnn_ref.adp_modify(cam[b, s], np.stack([origs[b * n_sc + s]] * 2), dataset, use_cls, x17) for s in range(n_sc))
            assert ref.shape == got[b].shape
            n_syn = Cout - len(use_cls)
            assert np.abs(got[b, :n_syn] - ref[:n_syn]).max() <= 2e-6, (dataset, b, np.abs(got[b] - ref).max())
            assert np.array_equal(got[b, n_syn:], ref[n_syn:])  # the use_cls channels: plain fp32 sums over the scales


def test_full_config_batch_properties(ctx):
    """BASELINE config 2/3 at full size (32 images = 64 samples at 321x321, M = 21, 10 iterations), checked through
    size-independent properties: an image's result does not depend on what else is in the batch (bit-exact, CNN
    and CRF), Q rows are distributions, labels are the arg-max of Q, and the run is bit-reproducible."""
    from wsscam.net import resnet50_cam

    B, S, C = 32, 321, 20
    sd = cnn_ref.make_resnet50_cam_state_dict(C, seed=0)
    rng = np.random.default_rng(77)
    imgs = [cnn_ref.synth_image(rng, S, S) for _ in range(B)]
    x = np.stack([cnn_ref.msf_pack(im, (S, S)) for im in imgs])
    cams = {}
    for precision in (_lib.PREC_F16, _lib.PREC_F16X3):  # the fast mode and the headline mode (its 256 x 256 tiles at 64 samples)
        m = resnet50_cam.CAM(None, "voc12", "", C, None, precision=precision)
        m.load_state_dict(sd)
        m.eval().cuda(0)
        cam = m.forward_batch(x)
        assert cam.shape == (B, C, 21, 21) and np.isfinite(cam).all() and (cam >= 0).all()
        for b in (0, 17, 31):
            assert np.array_equal(m.forward_batch(x[b:b + 1])[0], cam[b]), (precision, b)
        assert np.array_equal(m.forward_batch(x), cam)
        cams[precision] = cam
        del m
    # the two modes compute the same maps to the fast mode's stated tolerance; the headline mode against the oracle on one image
    nrm = lambda c: c / (c.max(axis=(2, 3), keepdims=True) + 1e-5)
    # (the fast mode's 2e-2 is stated on 3-4 images; the worst of 32 images x 20 noise-like maps is larger)
    d_modes = float(np.abs(nrm(cams[_lib.PREC_F16]) - nrm(cams[_lib.PREC_F16X3])).max())
    print("full batch: max |f16 - f16x3| on the normalised maps of 32 images = %.2e" % d_modes)
    assert d_modes <= 1e-1
    with torch.no_grad():
        rc = cnn_ref.resnet50_cam_forward(torch.from_numpy(x[5]), sd).numpy()
    assert np.abs(nrm(cams[_lib.PREC_F16X3][5:6])[0] - rc / (rc.max(axis=(1, 2), keepdims=True) + 1e-5)).max() <= 1e-4
    cam = cams[_lib.PREC_F16X3]

    M = C + 1
    hi = np.stack([np.stack([np.full((S, S), 0.15, np.float32)] + [cnn_ref.resize_bilinear_f64(
        (cam[b, c] / (cam[b, c].max() + 1e-5))[..., None], (S, S))[..., 0].astype(np.float32) for c in range(C)])
        for b in range(B)])
    p = hi / hi.sum(1, keepdims=True)
    U = np.ascontiguousarray(-np.log(np.clip(p, 1e-5, 1.0)).reshape(B, M, S * S).astype(np.float32))
    rgb = np.stack(imgs)
    cfg = (1.5, 3, 40, 13, 10, 10)

    def run(rgbs, Us):
        n = len(rgbs)
        crf = _lib.Crf(ctx, ctx.to_device(np.ascontiguousarray(rgbs)), n, S, S, cfg[0], cfg[2], cfg[3])
        q_dev, a_dev = ctx.alloc(n * M * S * S * 4), ctx.alloc(n * S * S * 4)
        crf.inference(ctx.to_device(np.ascontiguousarray(Us)), M, cfg[1], cfg[4], cfg[5], q_dev, a_dev)
        q, a = ctx.to_host(q_dev, (n, M, S * S), np.float32), ctx.to_host(a_dev, (n, S * S), np.int32)
        vg, vb = crf.lattice_sizes()
        crf.close()
        q_dev.free()
        a_dev.free()
        return q, a, vg, vb

    q, a, vg, vb = run(rgb, U)
    assert np.abs(q.sum(1) - 1).max() <= 1e-5 and (q >= 0).all()
    assert np.array_equal(a, q.argmax(1))
    assert len(set(vg)) == 1 and min(vb) > 1000
    for b in (0, 31):
        q1, a1, _, vb1 = run(rgb[b:b + 1], U[b:b + 1])
        assert np.array_equal(q1[0], q[b]) and np.array_equal(a1[0], a[b]) and vb1[0] == vb[b]
    q2, a2, _, _ = run(rgb, U)
    assert np.array_equal(q2, q) and np.array_equal(a2, a)


def _cv2_nearest(a, out_hw):
    """cv2.resize(a, (W, H), interpolation=cv2.INTER_NEAREST): source index min(floor(x * (1. / (dst / src))), src - 1)."""
    H, W = out_hw
    h, w = a.shape
    sy = np.minimum(np.floor(np.arange(H) * (1.0 / (H / h))).astype(np.int64), h - 1)
    sx = np.minimum(np.floor(np.arange(W) * (1.0 / (W / w))).astype(np.int64), w - 1)
    return a[sy][:, sx]


@pytest.mark.parametrize("dataset", ["adp_func", "deepglobe"])
def test_eval_cam_adp_deepglobe_branch(tmp_path, dataset):
    """eval_cam.run for the ADP / DeepGlobe branch (eval_cam.py:53-63, 89-115): keys[argmax(high_res | cam)] without a
    background channel, nearest-neighbour resize to the ground truth's size, confusion matrix, CSV rows (DeepGlobe drops its
    last class), colour PNGs -- against the numpy restatement; ground truth read from colour-coded PNGs as the dataset
    classes do (adp_semantic_segmentation_dataset.py:55-62)."""
    from PIL import Image

    from wsscam.step import eval_cam

    rng = np.random.default_rng(17)
    if dataset == "adp_func":
        names = {"bg": ["Background", "Other"], "fg": ["G.O", "G.N", "T"]}
        colours, field, out_hw, src_hws = eval_cam.ADP_CLS_COLOURS["func"], "high_res", (88, 88), [(22, 22), (30, 41), (88, 88)]
        label_dir = tmp_path / "SegmentationClassAug" / "ADP-func"
        split_file, n_eval = "segtest", 5
    else:
        names = {"bg": [], "fg": ["urban", "agriculture", "rangeland", "forest", "water", "unknown"]}
        colours, field, out_hw, src_hws = eval_cam.DEEPGLOBE_CLS_COLOURS, "cam", (96, 96), [(24, 24), (24, 24), (13, 31)]
        label_dir = tmp_path / "SegmentationClassAug"
        split_file, n_eval = "test", 5  # the 6th class never occurs: its row is dropped from the report
    n_class = len(names["bg"]) + len(names["fg"])
    os.makedirs(label_dir)
    os.makedirs(tmp_path / "ImageSets" / "Segmentation")
    os.makedirs(tmp_path / "cams")
    ids = ["img_%d" % i for i in range(len(src_hws))]
    (tmp_path / "ImageSets" / "Segmentation" / (split_file + ".txt")).write_text("\n".join(ids) + "\n")
    conf_ref = np.zeros((n_class, n_class), np.int64)
    preds_ref = []
    for i, (name, shw) in enumerate(zip(ids, src_hws)):
        gt = rng.integers(0, n_eval, out_hw)
        Image.fromarray(np.asarray(colours, np.uint8)[gt]).save(str(label_dir / (name + ".png")))
        keys = np.sort(rng.choice(n_eval, size=int(rng.integers(1, n_eval + 1)), replace=False)).astype(np.int64)
        maps = rng.random((len(keys),) + shw).astype(np.float32)
        maps[:, : shw[0] // 2] = np.round(maps[:, : shw[0] // 2], 1)  # ties: the first maximum wins
        np.save(str(tmp_path / "cams" / (name + ".npy")), {"keys": keys, field: maps, "cam" if field != "cam" else "x": maps[:1]})
        pred = _cv2_nearest(keys[np.argmax(maps, axis=0)], out_hw)
        preds_ref.append(pred)
        np.add.at(conf_ref, (gt.ravel(), pred.ravel()), 1)
    args = types.SimpleNamespace(dataset=dataset, chainer_eval_set="evaluation" if dataset == "adp_func" else "test",
                                 dev_root=str(tmp_path), cam_out_dir=str(tmp_path / "cams"), class_names=names,
                                 class_colours={"bg": [tuple(c) for c in colours[:len(names["bg"])]],
                                                "fg": [tuple(c) for c in colours[len(names["bg"]):]]},
                                 cam_clr_out_dir=str(tmp_path / "clr"), eval_dir=str(tmp_path / "eval"), run_name="run",
                                 split="evaluation", logfile=str(tmp_path / "log.txt"), overlay_r=0.75, cam_eval_thres=0.15)
    conf, s = eval_cam.run(args, batch_images=2)
    ref = conf_ref[:-1, :-1] if dataset == "deepglobe" else conf_ref
    assert np.array_equal(conf, ref)
    sr = eval_cam.scores_from_confusion(ref)
    assert np.allclose(s["iou"], sr["iou"], equal_nan=True) and np.isclose(s["miou"], sr["miou"])
    lines = open(os.path.join(args.eval_dir, "run_evaluation_cam_iou.csv")).read().splitlines()
    assert lines[0] == ",iou,precision,recall" and len(lines) == 1 + ref.shape[0] + 1 and lines[-1].startswith("mean,")
    for name, pred in zip(ids, preds_ref):
        clr = np.asarray(Image.open(os.path.join(args.cam_clr_out_dir, name + ".png")))
        assert np.array_equal(clr, np.asarray(colours, np.uint8)[pred])


def test_make_cam_native_sizes_outsize_none(tmp_path):
    """args.outsize = None -- the reference's resnet50 configuration (03b_irn/func_sample.py:143-148): no resize, every image
    goes through the network at its OWN, non-square size (wsc_net_forward_cam_hw; batches bucketed by size).  make_cam.run on
    images of four sizes (two share one), one .npy each, values vs the oracle's make_cam_image on the same native-size input,
    at the fp32-class bound."""
    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    rng = np.random.default_rng(12)
    sizes = [(75, 100), (100, 75), (75, 100), (66, 81), (97, 97)]
    labels = [np.zeros(20, np.float32) for _ in sizes]
    for i, cls in enumerate([[1, 4], [7], [2], [0, 19], [12]]):
        labels[i][cls] = 1
    data = [{"name": "2008_%06d" % i, "img": cnn_ref.msf_pack(cnn_ref.synth_image(rng, *sz), None), "size": sz, "label": lb}
            for i, (sz, lb) in enumerate(zip(sizes, labels))]
    assert data[0]["img"].shape == (2, 3, 75, 100)
    args = types.SimpleNamespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                                 use_cls=None, model_id="resnet50", cam_weights_name=None, state_dict=sd, outsize=None,
                                 dataset_obj=data, split="train_aug", cam_out_dir=str(tmp_path), n_gpus=1, cam_batch_images=4)
    make_cam.run(args)
    assert sorted(os.listdir(tmp_path)) == [d["name"] + ".npy" for d in data]
    for d in data:
        rec = np.load(os.path.join(tmp_path, d["name"] + ".npy"), allow_pickle=True).item()
        ref = cnn_ref.make_cam_image(torch.from_numpy(d["img"]), sd, d["size"], torch.from_numpy(d["label"]))
        assert np.array_equal(rec["keys"], ref["keys"])
        assert rec["cam"].shape == ref["cam"].shape and rec["high_res"].shape == ref["high_res"].shape
        assert np.abs(rec["high_res"] - ref["high_res"]).max() <= 1e-4, np.abs(rec["high_res"] - ref["high_res"]).max()
        assert np.abs(rec["cam"] - ref["cam"]).max() <= 1e-4


def test_round4_api_edges(ctx):
    """Error behaviour and corner inputs of the round-4 entry points: wsc_ctx_set_option, wsc_crf_v_*, wsc_label_confusion_nn,
    wsc_net_forward_cam_hw."""
    # unknown option
    with pytest.raises(_lib.WscError) as ei:
        ctx.set_option(99, 1)
    assert ei.value.status == _lib.WSC_ERR_INVALID
    # ragged CRF: a single 1-class image (Q = 1 everywhere), a zero-sized image is an error, M = 0 is an error
    rng = np.random.default_rng(3)
    from tests import helpers

    rgb, U, _ = helpers.synth_crf_case(rng, 9, 7, 1)
    cv = _lib.CrfV(ctx, [ctx.to_device(rgb)], [(9, 7)], 1.5, 40, 13)
    q_dev, a_dev = ctx.alloc(63 * 4), ctx.alloc(63 * 4)
    cv.inference([ctx.to_device(U)], [1], 3, 10, 3, [q_dev], [a_dev])
    assert np.allclose(ctx.to_host(q_dev, (63,), np.float32), 1.0) and not ctx.to_host(a_dev, (63,), np.int32).any()
    with pytest.raises(_lib.WscError):
        cv.inference([ctx.to_device(U)], [0], 3, 10, 3, [q_dev], [a_dev])
    with pytest.raises(_lib.WscError):
        cv.inference([ctx.to_device(U)], [1], 3, 10, 3, None, None)  # nothing to write
    cv.close()
    with pytest.raises(_lib.WscError):
        _lib.CrfV(ctx, [ctx.to_device(rgb)], [(0, 7)], 1.5, 40, 13)
    # label confusion: a label outside [0, n_class) is reported, not counted
    lab = np.zeros((1, 4, 4), np.int32)
    lab[0, 0, 0] = 7
    gt = np.zeros((1, 8, 8), np.uint8)
    conf = ctx.alloc(3 * 3 * 8)
    _lib.check(ctx._lib.wsc_memset(ctx.h, conf.ptr, 0, 72))
    with pytest.raises(_lib.WscError) as ei:
        _lib.label_confusion_nn(ctx, ctx.to_device(lab), [(4, 4)], [(8, 8)], [0], ctx.to_device(gt), 3, conf)
    assert ei.value.status == _lib.WSC_ERR_INVALID and "outside" in str(ei.value)
    # ignore_label pixels are skipped; everything else lands in one cell
    lab[0, 0, 0] = 1
    gt[0, :4, :4] = 255
    _lib.check(ctx._lib.wsc_memset(ctx.h, conf.ptr, 0, 72))
    _lib.label_confusion_nn(ctx, ctx.to_device(lab), [(4, 4)], [(8, 8)], [0], ctx.to_device(gt), 3, conf)
    c = ctx.to_host(conf, (3, 3), np.int64)
    assert c.sum() == 48 and c[0, 0] == 48  # the 2 x 2 block of label 1 lies under the ignored quadrant


@pytest.mark.parametrize("precision", [_lib.PREC_F16X3, _lib.PREC_F16])
def test_forward_cam_hw_matches_square_call_and_oracle(precision):
    """wsc_net_forward_cam_hw: on a square input it is the square entry point (same bits); on non-square inputs of odd sizes
    (65 x 97, 97 x 33) the CAM matches the oracle's forward at the mode's bound."""
    from wsscam.net import resnet50_cam

    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    m = resnet50_cam.CAM(None, "voc12", "", 20, None, precision=precision)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    ctx = m.ctx
    net = m._ensure_net()
    rng = np.random.default_rng(8)
    x = rng.normal(0, 1, (2, 2, 3, 64, 64)).astype(np.float32)
    h = net.cam_size(64)
    assert net.cam_size_hw(64, 64) == (h, h)
    c1, c2 = ctx.alloc(2 * 20 * h * h * 4), ctx.alloc(2 * 20 * h * h * 4)
    net.forward_cam(ctx.to_device(x), 2, 64, c1)
    net.forward_cam_hw(ctx.to_device(x), 2, 64, 64, c2)
    assert np.array_equal(ctx.to_host(c1, (2, 20, h, h), np.float32), ctx.to_host(c2, (2, 20, h, h), np.float32))
    tol = 2e-5 if precision == _lib.PREC_F16X3 else 5e-3
    for (H, W) in ((65, 97), (97, 33)):
        img = cnn_ref.synth_image(rng, H, W)
        xp = cnn_ref.msf_pack(img, None)
        hh, ww = net.cam_size_hw(H, W)
        cd = ctx.alloc(20 * hh * ww * 4)
        net.forward_cam_hw(ctx.to_device(xp[None]), 1, H, W, cd)
        cam = ctx.to_host(cd, (20, hh, ww), np.float32)
        with torch.no_grad():
            ref = cnn_ref.resnet50_cam_forward(torch.from_numpy(xp), sd).numpy()
        assert cam.shape == ref.shape, (cam.shape, ref.shape)
        assert np.abs(cam - ref).max() <= tol * ref.max(), (H, W, np.abs(cam - ref).max() / ref.max())


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(64, 64), (65, 97), (97, 33), (225, 225), (24, 24)])
def test_stem_pool_fused_equals_unfused(hw):
    """stem_pool_kernel (conv 7x7/2 + BN + ReLU + MaxPool 3/2/1 in one launch, resnet50.py:54-64) against the two-launch form
    it replaces (WSC_OPT_STEM_POOL_FUSED = 0): same accumulation sequence, same rounding to the hi/lo pair before the maximum,
    so the CAM of the whole network has the same bits -- on sizes whose last tile is partial in either direction."""
    from wsscam.net import resnet50_cam

    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=1)
    m = resnet50_cam.CAM(None, "voc12", "", 20, None, precision=_lib.PREC_F16X3)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    ctx = m.ctx
    net = m._ensure_net()
    H, W = hw
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.normal(0, 1, (3, 2, 3, H, W)).astype(np.float32)
    hh, ww = net.cam_size_hw(H, W)
    outs = []
    for fused in (1, 0):
        with ctx.option(_lib.OPT_STEM_POOL_FUSED, fused):
            cd = ctx.alloc(3 * 20 * hh * ww * 4)
            net.forward_cam_hw(ctx.to_device(x), 3, H, W, cd)
            outs.append(ctx.to_host(cd, (3, 20, hh, ww), np.float32))
    assert np.isfinite(outs[0]).all() and outs[0].max() > 0
    assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max() / outs[1].max()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [_lib.PREC_F16X3, _lib.PREC_F16])
def test_stage_entry_fusion_with_extreme_batchnorm_scales(precision):
    """The first block of a stage runs conv3 and the projection shortcut as one GEMM with both BatchNorm scales folded into the
    weights relative to sigma_c = max(|s3_c|, |sd_c|) (csrc/net.hip, resnet50.py:44-52).  BatchNorm parameters a real checkpoint
    can hold: a zero-initialised bn3 (gamma = 0: the residual branch is off), a zero shortcut scale, both zero, negative scales,
    and scales five orders of magnitude apart in the same channel -- the CAM must stay at the mode's bound against the oracle."""
    from wsscam.net import resnet50_cam

    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=4)
    for li in (1, 2, 3, 4):
        w3, wd = sd["resnet50.layer%d.0.bn3.weight" % li], sd["resnet50.layer%d.0.downsample.1.weight" % li]
        n = w3.numel()
        w3[0:n // 8] = 0.0                      # zero-initialised residual BatchNorm
        wd[n // 16:n // 8 + n // 16] = 0.0      # ... overlapping a zero shortcut scale: channels with both, and with either
        w3[n // 4:n // 4 + n // 8] *= -1.0      # negative scales
        wd[n // 2:n // 2 + n // 8] *= -1.0
        w3[3 * n // 4:3 * n // 4 + n // 16] *= 1e-5   # tiny next to the shortcut's
        wd[7 * n // 8:7 * n // 8 + n // 16] *= 1e-5   # and the other way round
    m = resnet50_cam.CAM(None, "voc12", "", 20, None, precision=precision)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    ctx = m.ctx
    net = m._ensure_net()
    rng = np.random.default_rng(11)
    S = 97
    img = cnn_ref.synth_image(rng, S, S)
    xp = cnn_ref.msf_pack(img, None)
    h = net.cam_size(S)
    cd = ctx.alloc(20 * h * h * 4)
    net.forward_cam(ctx.to_device(xp[None]), 1, S, cd)
    cam = ctx.to_host(cd, (20, h, h), np.float32)
    with torch.no_grad():
        ref = cnn_ref.resnet50_cam_forward(torch.from_numpy(xp), sd).numpy()
    assert np.isfinite(cam).all() and ref.max() > 0
    tol = 2e-5 if precision == _lib.PREC_F16X3 else 5e-3
    assert np.abs(cam - ref).max() <= tol * ref.max(), np.abs(cam - ref).max() / ref.max()
