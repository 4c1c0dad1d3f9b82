"""GPU: whole CAM networks + CAM tail through the C ABI against the torch fp32 oracle.

Stated tolerances on the per-class max-normalised CAM maps make_cam writes (values in [0,1]),
max over every pixel of every class of every image, vs the fp32 oracle:
  f16    (default product precision): max|d| <= 2e-2      (measured 1.5e-2, mean 3e-4)
  bf16   (fast mode)                : max|d| <= 1e-1      (measured 8.3e-2, mean 2e-3)
  bf16x3 (split bf16, 16-bit operands): max|d| <= 2e-4      (measured 1.5e-4)
  f16x3  (fp32-class mode, headline)  : max|d| <= 1e-4      (BASELINE.md section 4's bar for the FP32 mode; measured 4.4e-5 --
                                                           the torch-CPU oracle itself moves by 3e-5 with its reduction order)
and on the raw (un-normalised) CAM: f16 5e-3, bf16 3e-2, bf16x3 2e-4, f16x3 2e-5 (x max(cam); measured 3e-6).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cnn_ref
from tests import helpers
from wsscam import _lib
from wsscam.net import resnet50_cam, vgg16_cam, m7_cam

pytestmark = pytest.mark.gpu

TOL_NORM = {_lib.PREC_BF16: 1e-1, _lib.PREC_F16: 2e-2, _lib.PREC_BF16X3: 2e-4, _lib.PREC_F16X3: 1e-4}
TOL_RAW = {_lib.PREC_BF16: 3e-2, _lib.PREC_F16: 5e-3, _lib.PREC_BF16X3: 2e-4, _lib.PREC_F16X3: 2e-5}
PRECISIONS = [_lib.PREC_BF16, _lib.PREC_F16, _lib.PREC_BF16X3, _lib.PREC_F16X3]


@pytest.fixture(scope="module")
def resnet_sd():
    return cnn_ref.make_resnet50_cam_state_dict(20, seed=0)


def _model(cls, sd, C, precision):
    m = cls(None, "voc12", "", C, None, precision=precision)
    m.load_state_dict(sd)
    return m.eval().cuda(0)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("S", [64, 97])
def test_resnet50_cam_vs_golden(golden, resnet_sd, precision, S):
    """Fixtures produced by the reference's own net/resnet50.py (tests/golden)."""
    model = _model(resnet50_cam.CAM, resnet_sd, 20, precision)
    cam = model.forward(golden["x_S%d" % S])
    ref = golden["cam_S%d" % S]
    assert cam.shape == ref.shape
    assert np.abs(cam - ref).max() <= TOL_RAW[precision] * ref.max(), np.abs(cam - ref).max() / ref.max()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_resnet50_make_cam_321(golden, resnet_sd, precision):
    """A 3-image batch at 321x321 (different native sizes) through forward_cam + cam_postprocess vs the
    oracle's make_cam_image, including the torch bilinear + max-normalise tail."""
    model = _model(resnet50_cam.CAM, resnet_sd, 20, precision)
    rng = np.random.default_rng(3)
    sizes = [(375, 500), (500, 333), (281, 500)]
    imgs = [golden["img_321"]] + [cnn_ref.synth_image(rng, h, w) for (h, w) in sizes[1:]]
    labels = [np.zeros(20, np.float32) for _ in sizes]
    labels[0][[3, 11]] = 1
    labels[1][[0]] = 1
    labels[2][[5, 7, 19]] = 1
    packs = [{"name": "img%d" % i, "img": cnn_ref.msf_pack(im, (321, 321)), "size": sz, "label": lb}
             for i, (im, sz, lb) in enumerate(zip(imgs, sizes, labels))]

    class Args:
        split = "train_aug"
        dataset = "voc12"
        cam_out_dir = None

    from wsscam.step import make_cam

    outs = make_cam.process_batch(model, packs, Args, save=False)
    for p, o in zip(packs, outs):
        ref = cnn_ref.make_cam_image(torch.from_numpy(p["img"]), resnet_sd, p["size"], torch.from_numpy(p["label"]))
        assert np.array_equal(o["keys"], ref["keys"])
        assert o["cam"].shape == ref["cam"].shape and o["high_res"].shape == ref["high_res"].shape
        assert o["cam"].dtype == np.float32 and o["high_res"].dtype == np.float32
        d1 = np.abs(o["cam"] - ref["cam"]).max()
        d2 = np.abs(o["high_res"] - ref["high_res"]).max()
        assert d1 <= TOL_NORM[precision] and d2 <= TOL_NORM[precision], (d1, d2)
        # arg-max label agreement of the high-res maps (what eval_cam / cam_to_ir_label consume)
        if len(ref["keys"]) > 1:
            agree = (o["high_res"].argmax(0) == ref["high_res"].argmax(0)).mean()
            assert agree >= {_lib.PREC_BF16: 0.98, _lib.PREC_F16: 0.995, _lib.PREC_BF16X3: 0.9999, _lib.PREC_F16X3: 0.9999}[precision], agree


def test_cam_tail_exact_vs_torch(ctx, golden):
    """The tail alone, fed the fixture CAM: fp32 bilinear + max-normalise vs torch, both outputs."""
    cam = np.ascontiguousarray(golden["cam_321"])
    keys = [golden["tail_keys"].astype(np.int32)]
    cam_dev = ctx.to_device(cam[None])
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, 1, 20, 21, 21, [(375, 500)], keys)
    K, h4, w4, H0, W0 = shapes[0]
    s = ctx.to_host(s_dev, (K, h4, w4), np.float32)
    h = ctx.to_host(h_dev, (K, H0, W0), np.float32)
    assert np.abs(s - golden["tail_strided"]).max() <= 2e-6
    assert np.abs(h[:, ::25, :] - golden["tail_highres_rows"]).max() <= 2e-6
    assert np.allclose([h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum()], golden["tail_highres_sum"],
                       rtol=1e-5)


def test_cam_tail_edge_cases(ctx):
    """Empty key lists, K=1, sizes not multiples of 4/16, tiny images."""
    rng = np.random.default_rng(9)
    C, h, w = 7, 21, 21
    cam = np.maximum(rng.normal(0.5, 1.0, (4, C, h, w)), 0).astype(np.float32)
    cam[2, 4] = 0.0  # an all-zero class map: 0 / (0 + 1e-5) = 0
    sizes = [(17, 23), (1, 1), (64, 48), (333, 500)]
    keys = [[0, 6], [], [4], [1, 2, 3]]
    cam_dev = ctx.to_device(cam)
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, 4, C, h, w, sizes, keys)
    s_all = ctx.to_host(s_dev, (max(sum(k * a * b for k, a, b, _, _ in shapes), 1),), np.float32)
    h_all = ctx.to_host(h_dev, (max(sum(k * a * b for k, _, _, a, b in shapes), 1),), np.float32)
    for b in range(4):
        K, h4, w4, H0, W0 = shapes[b]
        assert K == len(keys[b])
        if K == 0:
            continue
        rs, rh = cnn_ref.make_cam_tail(torch.from_numpy(cam[b]), sizes[b], torch.tensor(keys[b]))
        s = s_all[s_off[b]:s_off[b] + K * h4 * w4].reshape(K, h4, w4)
        hh = h_all[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0)
        assert np.abs(s - rs.numpy()).max() <= 2e-6 and np.abs(hh - rh.numpy()).max() <= 2e-6


def test_cam_tail_signed_maps(ctx):
    """The ADP 'func' background channel is bg - max(exception CAMs) with no ReLU (common_cam.py:57-75): on an all-tissue
    patch it is <= 0 everywhere and the reference divides by (negative max + 1e-5).  Maps of either sign and an
    all-negative map against torch (make_cam.py:71-76), through both the tail and the fused unary path's maximum."""
    rng = np.random.default_rng(19)
    C, h, w = 5, 14, 14
    cam = rng.normal(0.0, 1.0, (2, C, h, w)).astype(np.float32)
    cam[0, 0] = -np.abs(cam[0, 0]) - 0.05      # all negative
    cam[1, 3] = -0.75                           # constant negative
    sizes = [(50, 37), (33, 64)]
    keys = [[0, 1, 4], [3, 2]]
    cam_dev = ctx.to_device(cam)
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, 2, C, h, w, sizes, keys)
    s_all = ctx.to_host(s_dev, (sum(k * a * b for k, a, b, _, _ in shapes),), np.float32)
    h_all = ctx.to_host(h_dev, (sum(k * a * b for k, _, _, a, b in shapes),), np.float32)
    for b in range(2):
        K, h4, w4, H0, W0 = shapes[b]
        rs, rh = cnn_ref.make_cam_tail(torch.from_numpy(cam[b]), sizes[b], torch.tensor(keys[b]))
        s = s_all[s_off[b]:s_off[b] + K * h4 * w4].reshape(K, h4, w4)
        hh = h_all[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0)
        tol = 2e-6 * max(1.0, float(np.abs(rh.numpy()).max()))
        assert np.abs(s - rs.numpy()).max() <= tol and np.abs(hh - rh.numpy()).max() <= tol
    # all-negative map / (negative max + 1e-5): every value is >= 1 (the old unsigned max divided by 1e-5 instead)
    assert 1.0 - 1e-4 <= h_all[h_off[0]:h_off[0] + 50 * 37].min() and h_all[h_off[0]:h_off[0] + 50 * 37].max() < 1e3


def test_bilinear_resize_vs_torch(ctx):
    rng = np.random.default_rng(2)
    src = rng.normal(0, 1, (5, 40, 40)).astype(np.float32)
    for (H, W) in [(41, 41), (321, 321), (20, 27)]:
        dst = ctx.alloc(5 * H * W * 4)
        _lib.bilinear_resize(ctx, ctx.to_device(src), 5, 40, 40, dst, H, W)
        out = ctx.to_host(dst, (5, H, W), np.float32)
        ref = torch.nn.functional.interpolate(torch.from_numpy(src)[None], (H, W), mode="bilinear",
                                              align_corners=False)[0].numpy()
        assert np.abs(out - ref).max() <= 2e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("batchnorm", [True, False])
def test_vgg16_cam(precision, batchnorm):
    """Modified VGG16 (conv(bias) -> ReLU -> BN, 2x2 pools), classifier scores and CAM, S=65."""
    C = 20
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, batchnorm, seed=1)
    model = _model(vgg16_cam.CAM, sd, C, precision)
    rng = np.random.default_rng(4)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 80, 90), (65, 65))
    cam, score = model.forward_batch(x[None], want_score=True)
    with torch.no_grad():
        rcam, rscore = cnn_ref.vgg16_cam_forward(torch.from_numpy(x), sd, C)
    assert cam[0].shape == tuple(rcam.shape) == (C, 8, 8)  # 65 -> 32 -> 16 -> 8 (three 2x2 pools)
    assert np.abs(cam[0] - rcam.numpy()).max() <= TOL_RAW[precision] * float(rcam.max())
    assert np.abs(score[0] - rscore.numpy()).max() <= {_lib.PREC_BF16: 5e-3, _lib.PREC_F16: 1e-3,
                                                       _lib.PREC_BF16X3: 1e-4, _lib.PREC_F16X3: 2e-5}[precision]


@pytest.mark.parametrize("precision,tol", [(_lib.PREC_F16X3, 1e-4), (_lib.PREC_BF16X3, 2e-4)])
def test_m7_cam(precision, tol):
    C = 20
    sd = cnn_ref.make_plain_state_dict("m7", cnn_ref.M7_CFG, C, True, seed=2)
    alpha = cnn_ref.grad_cam_weights(sd, "m7", cnn_ref.M7_CFG, 32, C)  # (F, C), 02_cues/utilities.py:60-99
    sd_dev = dict(sd)
    sd_dev["gradcam_weights"] = torch.from_numpy(alpha.astype(np.float32))
    model = _model(m7_cam.CAM, sd_dev, C, precision)
    rng = np.random.default_rng(5)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 70, 60), (64, 64))
    cam, score = model.forward_batch(x[None], want_score=True)
    with torch.no_grad():
        rcam, rscore = cnn_ref.m7_cam_forward(torch.from_numpy(x), sd, torch.from_numpy(alpha), C)
    assert cam[0].shape == tuple(rcam.shape) == (C, 16, 16)
    assert np.abs(cam[0] - rcam.numpy()).max() <= tol * max(float(rcam.max()), 1e-3)
    assert np.abs(score[0] - rscore.numpy()).max() <= tol


def test_state_dict_errors(ctx, resnet_sd):
    sd = {k: v.numpy() for k, v in resnet_sd.items()}
    bad = dict(sd)
    del bad["resnet50.layer2.0.bn2.running_var"]
    with pytest.raises(_lib.WscError) as ei:
        _lib.Net(ctx, _lib.ARCH_RESNET50_CAM, bad, 20)
    assert ei.value.status == _lib.WSC_ERR_MISSING_KEY and "running_var" in str(ei.value)
    bad = dict(sd)
    bad["classifier.weight"] = bad["classifier.weight"][:10]
    with pytest.raises(_lib.WscError) as ei:
        _lib.Net(ctx, _lib.ARCH_RESNET50_CAM, bad, 20)
    assert ei.value.status == _lib.WSC_ERR_SHAPE


# ---- 02_cues / 03c_hsn Grad-CAM mirrors -------------------------------------------------------------
def _vgg_model(C, seed):
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, True, seed=seed)
    return _model(vgg16_cam.CAM, sd, C, _lib.PREC_F16X3), sd  # the package default / headline mode


def test_get_grad_cam_weights_closed_form_vs_autograd():
    """02_cues/utilities.py:60-99 on the GAP+Linear VGG16: closed form == torch autograd on a zeros image."""
    from wsscam.cues import utilities as cues

    C = 20
    model, sd = _vgg_model(C, seed=3)
    alpha = cues.get_grad_cam_weights(model, cues.find_final_layer(model), np.zeros((1, 65, 65, 3), np.float32))
    ref = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 65, C)
    assert alpha.shape == ref.shape == (1024, C)
    assert np.abs(alpha - ref).max() <= 1e-4 * np.abs(ref).max()


def test_cues_grad_cam_and_resize_stack():
    """02_cues/utilities.py:101-144 (einsum + relu, keep_inds, threshold mask) and :20-40 (resize_stack)."""
    from wsscam.cues import utilities as cues

    C = 20
    model, sd = _vgg_model(C, seed=4)
    rng = np.random.default_rng(6)
    from oracle import hsn_ref

    # (B, S, S, 3) NHWC as read_batch gives it: the uint8 batch of cv2.resize, normalised (02_cues/utilities.py:172-180)
    imgs = cnn_ref.normalize_int(hsn_ref.read_batch_u8([cnn_ref.synth_image(rng, 70, 80) for _ in range(3)], (65, 65)).astype(np.float64))
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 65, C)
    keep = np.array([0, 2, 5, 19])
    is_pass = rng.random((3, len(keep))) > 0.3
    out = cues.grad_cam(model, alpha, imgs, is_pass, "x", keep)
    # conv_val = output of the final Activation (pre-BatchNorm): 02_cues/utilities.py:129-132
    conv_val = cnn_ref.keras_conv_val(torch.from_numpy(np.transpose(imgs, (0, 3, 1, 2)).copy()), sd, "vgg16",
                                      cnn_ref.VGG16_CFG).numpy().astype(np.float64)
    ref = np.maximum(np.einsum("ijkl,lm->ijkm", conv_val, alpha), 0)[:, :, :, keep] * is_pass[:, None, None, :]
    assert out.shape == ref.shape == (3, 8, 8, 4)
    assert np.abs(out - ref).max() <= 1e-4 * ref.max()  # f16x3, the package default
    # resize_stack to the 41x41 seed size vs torch bilinear (cv2 INTER_LINEAR has the same half-pixel centres)
    st = np.transpose(ref, (0, 3, 1, 2))
    rs = cues.resize_stack(st, (41, 41), ctx=model.ctx)
    tr = torch.nn.functional.interpolate(torch.from_numpy(st), (41, 41), mode="bilinear", align_corners=False).numpy()
    assert rs.shape == (3, 4, 41, 41) and np.abs(rs - tr).max() <= 1e-5 * max(tr.max(), 1)


def test_hsn_grad_cam_and_postprocessing():
    """03c_hsn/utilities.py:231-278 (grad_cam), :306-364 (modify_by_htt), :367-397 (get_cs_gradcam)."""
    from wsscam.hsn import utilities as hsn

    C = 20
    model, sd = _vgg_model(C, seed=5)
    rng = np.random.default_rng(7)
    from oracle import hsn_ref

    raw = [cnn_ref.synth_image(rng, 70, 80) for _ in range(2)]
    imgs = cnn_ref.normalize_int(hsn_ref.read_batch_u8(raw, (65, 65)).astype(np.float64))
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 65, C)
    scores = rng.random((2, C))
    is_pass = scores > 0.4
    out = hsn.grad_cam(model, alpha, imgs, is_pass, "x", scores, orig_sz=[65, 65], should_upsample=True)
    conv_val = cnn_ref.keras_conv_val(torch.from_numpy(np.transpose(imgs, (0, 3, 1, 2)).copy()), sd, "vgg16",
                                      cnn_ref.VGG16_CFG).numpy().astype(np.float64)  # final Activation, pre-BatchNorm
    cams = np.einsum("ijkl,lm->ijkm", conv_val, alpha)
    up = torch.nn.functional.interpolate(torch.from_numpy(np.transpose(cams, (0, 3, 1, 2))), (65, 65), mode="bilinear",
                                         align_corners=False).numpy()
    up = np.maximum(np.transpose(up, (0, 2, 3, 1)), 0)
    ref = up / np.maximum(up.max(axis=(1, 2, 3), keepdims=True), 1e-7) * (scores * is_pass)[:, None, None, :]
    assert out.shape == ref.shape == (2, 65, 65, C)
    assert np.abs(out - ref).max() <= 1e-4  # maps normalised to a maximum of 1; f16x3

    # get_cs_gradcam: margin on the arg-max class only; 'Other' passes through for func
    g = rng.random((2, 4, 5, 6))
    classes = ["Background", "Other", "G.O", "T"]
    cs = hsn.get_cs_gradcam(g.copy(), classes, "func")
    srt = np.sort(g, axis=1)
    for c in range(4):
        exp = g[:, c] if c == 1 else (srt[:, -1] - srt[:, -2]) * (g.argmax(1) == c)
        assert np.allclose(cs[:, c], exp)
    # modify_by_htt (morph): background = smoothed 0.75*sigmoid(4(mean-240)) minus max exception CAM
    import scipy.ndimage
    import scipy.special
    classes_m = ["Background", "E", "A.W", "A.B", "A.M"]
    gm = rng.random((2, 5, 9, 9))
    ims = rng.integers(200, 256, (2, 9, 9, 3)).astype(np.float64)
    outm = hsn.modify_by_htt(gm.copy(), ims, classes_m)
    bg = np.stack([scipy.ndimage.gaussian_filter(0.75 * scipy.special.expit(4 * (ims[i].mean(-1) - 240)), sigma=2)
                   for i in range(2)])
    assert np.allclose(outm[:, 0], bg - gm[:, 2:5].max(1))
    assert np.array_equal(outm[:, 1:], gm[:, 1:])


def test_crf_inference_mirror(ctx):
    """lib.crf.crf_inference (03a_sec-dsrg call sites): marginals (H, W, C) vs the C oracle."""
    from tests import helpers
    from wsscam.misc import imutils

    rng = np.random.default_rng(15)
    H, W, C = 41, 41, 5
    rgb, _, p = helpers.synth_crf_case(rng, H, W, C)
    fm = np.transpose(p, (1, 2, 0)).astype(np.float32)
    cfg = {"g_sxy": 3 / 12, "g_compat": 3, "bi_sxy": 80 / 12, "bi_srgb": 13, "bi_compat": 10, "iterations": 5}
    out = imutils.crf_inference(rgb, cfg, C, fm, use_log=True, ctx=ctx)
    U = np.ascontiguousarray(-np.log(np.transpose(fm, (2, 0, 1)).reshape(C, -1)))
    qr, _, _ = helpers.crf_oracle(rgb, U, (cfg["g_sxy"], 3, cfg["bi_sxy"], 13, 10, 5))
    assert out.shape == (H, W, C) and out.dtype == np.float32
    assert np.abs(out - np.transpose(qr.reshape(C, H, W), (1, 2, 0))).max() <= 1e-3


def test_eval_cam_confusion_exact(ctx, tmp_path):
    """eval_cam on the device (eval_cam.py:48-62 + chainercv confusion): integer work, bit-exact vs numpy."""
    from wsscam.step import eval_cam

    rng = np.random.default_rng(21)
    C, h = 20, 21
    cam = np.maximum(rng.normal(0.3, 1.0, (3, C, h, h)), 0).astype(np.float32)
    sizes = [(37, 50), (64, 41), (1, 7)]
    keys = [[2, 9, 19], [], [0]]
    cam_dev = ctx.to_device(cam)
    s_dev, h_dev, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, 3, C, h, h, sizes, keys)
    gts = [rng.integers(0, 21, sz).astype(np.uint8) for sz in sizes]
    for g in gts:
        g[rng.random(g.shape) < 0.1] = 255  # ignore label
    acc = eval_cam.ConfusionAccumulator(ctx, n_class=21, cam_eval_thres=0.15)
    preds = acc.add_batch(h_dev, sizes, keys, h_off, gts, want_pred=True)
    acc.add_batch(h_dev, sizes, keys, h_off, gts)  # accumulates: second pass doubles the counts
    conf = acc.confusion()
    hi_all = ctx.to_host(h_dev, (max(sum(k * a * b for k, _, _, a, b in shapes), 1),), np.float32)
    ref = np.zeros((21, 21), np.int64)
    for b, (sz, ks, gt) in enumerate(zip(sizes, keys, gts)):
        K, _, _, H0, W0 = shapes[b]
        hr = hi_all[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0)
        cams = np.pad(hr, ((1, 0), (0, 0), (0, 0)), mode="constant", constant_values=0.15)  # eval_cam.py:50
        kk = np.pad(np.asarray(ks, dtype=np.int64) + 1, (1, 0), mode="constant")              # eval_cam.py:51
        cls = kk[np.argmax(cams, axis=0)]
        assert np.array_equal(preds[b], cls.astype(np.uint8))
        m = gt != 255
        ref += np.bincount(21 * gt[m].astype(np.int64) + cls[m], minlength=441).reshape(21, 21)
    assert np.array_equal(conf, 2 * ref)
    s = eval_cam.scores_from_confusion(conf)
    assert 0 <= s["miou"] <= 1

    class Args:
        eval_dir = str(tmp_path); run_name = "r"; split = "val"; logfile = str(tmp_path / "log.txt")

    eval_cam.write_report(Args, conf, ["c%d" % i for i in range(21)])
    assert "[eval_cam, val] miou: " in open(Args.logfile).read()
    assert open(tmp_path / "r_val_cam_iou.csv").readline().strip() == ",iou,precision,recall"


def test_gen_cues_driver(tmp_path):
    """02_cues/demo.py:26-222 for VOC2012: fg + bg models, class gating by score threshold x image label, Grad-CAM ->
    41 x 41 seeds -> localization_cues.pickle; every stage against its torch / numpy restatement."""
    import pickle

    from wsscam.cues import demo as cues_demo
    from wsscam.cues import utilities as cues

    C = 20
    fg, sd_fg = _vgg_model(C, seed=8)
    bg, sd_bg = _vgg_model(C, seed=9)
    rng = np.random.default_rng(10)
    # two images need resizing (read_batch: cv2.resize to a uint8 batch, 02_cues/utilities.py:172-176), one is at the network size
    images = [cnn_ref.synth_image(rng, 300, 340), cnn_ref.synth_image(rng, 375, 500), cnn_ref.synth_image(rng, 321, 321)]
    labels = (rng.random((3, C)) < 0.2).astype(np.float64)
    labels[:, 3] = 1
    alphas = {"fg": cnn_ref.grad_cam_weights(sd_fg, "vgg16", cnn_ref.VGG16_CFG, 33, C),
              "bg": cnn_ref.grad_cam_weights(sd_bg, "vgg16", cnn_ref.VGG16_CFG, 33, C)}
    thr = {"fg": np.full((1, C), 0.45), "bg": np.full((1, C), 0.45)}
    out = cues_demo.gen_cues("VOC2012", "VGG16", 0.2, 2, models={"fg": fg, "bg": bg}, alphas=alphas, thresholds=thr,
                             images=images, labels=labels, out_dir=str(tmp_path), is_verbose=False)
    saved = pickle.load(open(tmp_path / "localization_cues.pickle", "rb"))
    assert sorted(saved) == sorted(out) == sorted(["%d_%s" % (i, k) for i in range(3) for k in ("cues", "labels")])
    # restatement with the oracle network (batch maxima are per batch of 2: Q7); the ORACLE quantises the batch itself
    # (oracle/hsn_ref.py's loop statement of OpenCV's 8-bit resize), so a driver that normalised an un-rounded resize would
    # feed the network inputs up to 1/255 away and move the thresholded seeds
    from oracle import hsn_ref

    batch_u8 = hsn_ref.read_batch_u8(images, (321, 321))
    assert batch_u8.dtype == np.uint8
    x = np.stack([cnn_ref.normalize_int(im.astype(np.float64)) for im in batch_u8])
    ref = {}
    for lo, hi in ((0, 2), (2, 3)):
        Hm, ip = {}, {}
        for m, sd, a in (("fg", sd_fg, alphas["fg"]), ("bg", sd_bg, alphas["bg"])):
            xt = torch.from_numpy(np.transpose(x[lo:hi], (0, 3, 1, 2)).astype(np.float32).copy())
            with torch.no_grad():
                feat, pre = cnn_ref.plain_features(xt, sd, "vgg16", cnn_ref.VGG16_CFG, return_pre_bn=True)
                sc = torch.sigmoid(torch.nn.functional.linear(feat.mean((2, 3)), sd["vgg16.classifier.0.weight"],
                                                              sd["vgg16.classifier.0.bias"])).numpy()
            ip[m] = (sc >= 0.45) * labels[lo:hi]
            cam = np.maximum(np.einsum("ijkl,lm->ijkm", np.transpose(pre.numpy(), (0, 2, 3, 1)).astype(np.float64), a), 0)
            cam = cam * ip[m][:, None, None, :]
            Hm[m] = torch.nn.functional.interpolate(torch.from_numpy(np.transpose(cam, (0, 3, 1, 2))), (41, 41),
                                                    mode="bilinear", align_corners=False).numpy()
        ci = [np.where(ip["fg"][i])[0] + 1 for i in range(hi - lo)]
        cues.get_fgbg_cues(ref, Hm["fg"], Hm["bg"], ci, list(range(lo, hi)), 0.2)
    for i in range(3):
        assert np.array_equal(out["%d_labels" % i], ref["%d_labels" % i])
        a, b = out["%d_cues" % i], ref["%d_cues" % i]
        la, lb = np.zeros((41, 41), np.int64), np.zeros((41, 41), np.int64)
        la[a[1], a[2]] = a[0] + 1
        lb[b[1], b[2]] = b[0] + 1
        assert (la == lb).mean() >= 0.995  # thresholded maps: a pixel on the 0.2 x max contour may flip


@pytest.mark.parametrize("quirk,resized", [(False, False), (True, False), (False, True), (True, True)])
def test_hsn_segment_driver(quirk, resized):
    """03c_hsn/demo.py:18-268 (VOC2012 branch) end to end at the real 321 x 321 size: scores -> 1/3 threshold ->
    HSN Grad-CAM -> batch-max background channel -> dense CRF; labels vs the oracle chain (torch net, numpy
    post-processing, C CRF).  quirk: the reference's actual VOC normalisation (utilities.py:142-146: uint8 wrap-around on
    image columns 0..2, in place, so the CRF sees the modified image too) instead of the intended per-channel one."""
    import scipy.special

    from wsscam.hsn import demo as hsn_demo

    C = 5
    sd_fg = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, True, seed=11)
    sd_bg = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, True, seed=12)
    fg = _model(vgg16_cam.CAM, sd_fg, C, _lib.PREC_F16X3)
    bg = _model(vgg16_cam.CAM, sd_bg, C, _lib.PREC_F16X3)
    rng = np.random.default_rng(13)
    if resized:  # read_batch's uint8 contract (03c_hsn/utilities.py:170-181): cv2.resize into a uint8 batch, on the path
        images = [cnn_ref.synth_image(rng, 375, 500), cnn_ref.synth_image(rng, 300, 290)]
    else:
        images = [cnn_ref.synth_image(rng, 321, 321) for _ in range(2)]
    alphas = {"fg": cnn_ref.grad_cam_weights(sd_fg, "vgg16", cnn_ref.VGG16_CFG, 33, C),
              "bg": cnn_ref.grad_cam_weights(sd_bg, "vgg16", cnn_ref.VGG16_CFG, 33, C)}
    out = hsn_demo.segment("VOC2012", "VGG16", 2, models={"fg": fg, "bg": bg}, alphas=alphas, images=[im.copy() for im in images],
                           is_verbose=False, reference_normalize_quirk=quirk)
    assert len(out) == 2 and out[0].shape == (321, 321)
    if resized:  # the oracle chain quantises on its own: the loop statement of OpenCV's 8-bit INTER_LINEAR
        from oracle import hsn_ref

        images = list(hsn_ref.read_batch_u8(images, (321, 321)))
    if quirk:
        batch = np.stack(images)          # normalize('VOC2012', img_batch) of the reference, verbatim
        batch[:, :, 0] -= 104
        batch[:, :, 1] -= 117
        batch[:, :, 2] -= 123
        x = batch / 255
        images = list(batch)              # in place: dcrf_process gets the modified batch (demo.py:111,167)
    else:
        x = np.stack([cnn_ref.normalize_int(im.astype(np.float64)) for im in images])
    xt = torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).astype(np.float32).copy())
    H = {}
    for m, sd in (("fg", sd_fg), ("bg", sd_bg)):
        with torch.no_grad():
            feat, pre = cnn_ref.plain_features(xt, sd, "vgg16", cnn_ref.VGG16_CFG, return_pre_bn=True)
            sc = torch.sigmoid(torch.nn.functional.linear(feat.mean((2, 3)), sd["vgg16.classifier.0.weight"],
                                                          sd["vgg16.classifier.0.bias"])).numpy().astype(np.float64)
        cams = np.einsum("ijkl,lm->ijkm", np.transpose(pre.numpy(), (0, 2, 3, 1)).astype(np.float64), alphas[m])
        up = torch.nn.functional.interpolate(torch.from_numpy(np.transpose(cams, (0, 3, 1, 2))), (321, 321),
                                             mode="bilinear", align_corners=False).numpy()
        up = np.maximum(up, 0)
        H[m] = up / np.maximum(up.max(axis=(1, 2, 3), keepdims=True), 1e-7) * (sc * (sc >= 1 / 3))[:, :, None, None]
    Y = np.zeros((2, C + 1, 321, 321))
    X_bg = H["bg"].sum(1)
    Y[:, 0] = 0.15 * scipy.special.expit(X_bg.max() - X_bg)
    Y[:, 1:] = H["fg"]
    cfg = (1.5, 3, 40, 13, 10, 10)
    for b in range(2):
        keep = np.where(Y[b].sum(axis=(1, 2)) > 0)[0]  # dcrf_process: classes with positive mass (utilities.py:425)
        p = Y[b][keep]
        U = np.ascontiguousarray(-np.log(np.clip(p, 1e-5, 1.0)).reshape(len(keep), -1).astype(np.float32))
        _, ar, _ = helpers_crf(images[b], U, cfg)
        ref = keep[ar.reshape(321, 321)]
        assert (out[b] == ref).mean() >= 0.99, (out[b] == ref).mean()


def helpers_crf(rgb, U, cfg):
    from tests import helpers

    return helpers.crf_oracle(rgb, U, cfg)


@pytest.mark.parametrize("precision,min_agree", [(_lib.PREC_F16X3, 0.99), (_lib.PREC_BF16X3, 0.99), (_lib.PREC_F16, 0.97)])
def test_hsn_segment_adp_driver(precision, min_agree):
    """03c_hsn/demo.py:271-380 (ADP, BASELINE config 5 without the 1088 x 1088 evaluation upsample): one 31-class
    VGG16 (no BatchNorm), Grad-CAM at 321 x 321, morph (29 classes) and func (5 classes) stacks with synthesised
    Background / Other channels, class-specific Grad-CAM, dense CRF per type; FINAL labels vs the all-fp32 oracle chain, in
    the fp32-class modes and in the fast half-precision mode (`bench.py --workload hsn --precision f16`), each with its
    stated per-image label agreement."""
    import scipy.ndimage
    import scipy.special

    from tests.test_gpu_edge import _adp_like_image
    from wsscam.hsn import demo as hsn_demo
    from wsscam.hsn import utilities as hsn

    C, S = 31, 321
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=21)
    model = _model(vgg16_cam.CAM, sd, C, precision)
    rng = np.random.default_rng(22)
    images = [_adp_like_image(rng, S, S) for _ in range(2)]
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    thr = np.full((1, C), 0.5)
    cfgs = {"morph": np.array([3 / 2, 3, 80 / 2, 13, 10, 5]), "func": np.array([3, 3, 50, 5, 10, 5])}
    out = hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S, 2)
    assert len(out["morph"]) == len(out["func"]) == 2 and out["morph"][0].shape == (S, S)

    ref = helpers.oracle_chain_hsn_adp(images, sd, alpha, 0.5, cfgs)
    for htt in ("morph", "func"):
        for b in range(2):
            agree = (out[htt][b] == ref[htt][b]).mean()
            print("hsn adp chain precision %d %s image %d: label agreement %.5f" % (precision, htt, b, agree))
            assert agree >= min_agree, (htt, b, agree)


def test_hsn_segment_adp_batches_in_flight_equal_serial():
    """Round 6: segment_adp keeps several batches in flight, each on its own stream / buffer pool / thread
    (hsn.demo.run_batches_on_lanes).  A batch's label maps must not depend on the lane it ran on nor on what ran beside it:
    five patches in batches of two on three lanes == the same call on one lane (the reference's serial loop), bit for bit,
    and the per-image CRF class counts come back in image order."""
    from tests.test_gpu_edge import _adp_like_image
    from wsscam.hsn import demo as hsn_demo

    C, S = 31, 129
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=23)
    model = _model(vgg16_cam.CAM, sd, C, _lib.PREC_F16X3)
    rng = np.random.default_rng(24)
    images = [_adp_like_image(rng, S, S) for _ in range(5)]
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    thr = np.full((1, C), 0.5)
    cfgs = {"morph": np.array([3 / 2, 3, 80 / 2, 13, 10, 3]), "func": np.array([3, 3, 50, 5, 10, 3])}
    st1, st3 = {}, {}
    serial = hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S, 2, stats=st1, n_lanes=1)
    # (twice: the second call finds the lanes' contexts, pools and cached Gaussian lattices warm; with chain_stacks the lanes'
    # VGG16 passes take turns on the device -- _lib.StackChain -- and the maps are the same)
    for chain in (False, True):
        st3 = {}
        lanes = hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S, 2, stats=st3, n_lanes=3, chain_stacks=chain)
        for htt in ("morph", "func"):
            assert len(lanes[htt]) == 5
            for b in range(5):
                assert np.array_equal(lanes[htt][b], serial[htt][b]), (htt, b)
        assert st1 == st3


def test_hsn_adp_driver_resized_patches_and_adipose_indexing():
    """segment_adp on 272 x 272 patches at a 224 network size: (1) ADPCues.read_batch's uint8 contract (adp_cues.py:122-128:
    cv2.resize into a uint8 batch -- the oracle quantises with its own loop statement of OpenCV's 8-bit rule); (2) the
    reference's adipose indexing (demo.py:368-369: positions in classes['morph'] applied to the valid stack = S.R, A.W, A.B;
    the bookkeeping itself is pinned on the host by tests/test_cues_host.py::test_adp_adipose_channels_follow_the_reference).
    S.R is forced to pass and the true adipose classes to fail, on patches that are tissue everywhere, so that the functional
    'Other' channel is where the two indexings differ most (printed)."""
    from tests.test_gpu_edge import _adp_like_image
    from wsscam.hsn import demo as hsn_demo

    C, S = 31, 224
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=23)
    model = _model(vgg16_cam.CAM, sd, C, _lib.PREC_F16X3)
    rng = np.random.default_rng(24)
    images = [cnn_ref.synth_image(rng, 272, 272), _adp_like_image(rng, 272, 272)]
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    thr = np.full((1, C), 0.5)
    thr[0, 17], thr[0, 18:21] = 0.0, 2.0  # S.R always passes, A.W / A.B / A.M never
    cfgs = {"morph": np.array([3 / 2, 3, 80 / 2, 13, 10, 3]), "func": np.array([3, 3, 50, 5, 10, 3])}
    out = hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S, 2)
    ref = helpers.oracle_chain_hsn_adp(images, sd, alpha, thr, cfgs, size=S)
    other = helpers.oracle_chain_hsn_adp(images, sd, alpha, thr, cfgs, size=S, adipose_as_written=False)
    for htt in ("morph", "func"):
        for b in range(2):
            agree = (out[htt][b] == ref[htt][b]).mean()
            print("hsn adp resized %s image %d: label agreement %.5f" % (htt, b, agree))
            assert agree >= 0.99, (htt, b, agree)
    sep = min((other["func"][b] == ref["func"][b]).mean() for b in range(2))
    print("hsn adp: reference vs intended adipose indexing agree on %.4f of the func labels" % sep)


def test_gen_cues_adp_driver(tmp_path):
    """02_cues/demo.py:224-310: ADP seeds for both HTT types from one 31-class model; checks the cue layout, that
    every cue class is a passing class (plus the synthesised Background / Other), and the restated chain for morph."""
    import pickle

    import scipy.ndimage
    import scipy.special

    from tests.test_gpu_edge import _adp_like_image
    from wsscam.cues import demo as cues_demo
    from oracle import hsn_ref
    from wsscam.cues import utilities as cues

    C, S = 31, 224
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=31)
    model = _model(vgg16_cam.CAM, sd, C, _lib.PREC_F16X3)
    rng = np.random.default_rng(32)
    # ADP patches are 272 x 272 and the network runs at 224: ADPCues.read_batch resizes into a uint8 batch (adp_cues.py:122-128)
    images = [_adp_like_image(rng, 272, 272), _adp_like_image(rng, 272, 272), _adp_like_image(rng, S, S)]
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    dirs = {"morph": str(tmp_path / "m"), "func": str(tmp_path / "f")}
    out = cues_demo.gen_cues_adp("VGG16", 0.2, 2, S, dirs, "tuning", False, model=model, alpha=alpha,
                                 thresholds=np.full((1, C), 0.5), images=images)
    assert pickle.load(open(tmp_path / "m" / "localization_cues.pickle", "rb")).keys() == out["morph"].keys()
    raw = hsn_ref.read_batch_u8(images, (S, S))  # the oracle's own uint8 batch
    xt = torch.from_numpy(np.transpose((raw - 193.09203) / 56.450138, (0, 3, 1, 2)).astype(np.float32).copy())
    with torch.no_grad():
        feat = cnn_ref.plain_features(xt, sd, "vgg16", cnn_ref.VGG16_CFG)
        sc = torch.sigmoid(torch.nn.functional.linear(feat.mean((2, 3)), sd["vgg16.classifier.0.weight"],
                                                      sd["vgg16.classifier.0.bias"])).numpy()
    ip = sc >= 0.5
    cam = np.maximum(np.einsum("ijkl,lm->ijkm", np.transpose(feat.numpy(), (0, 2, 3, 1)).astype(np.float64), alpha), 0)
    H = torch.nn.functional.interpolate(torch.from_numpy(np.transpose(cam * ip[:, None, None, :], (0, 3, 1, 2))), (41, 41),
                                        mode="bilinear", align_corners=False).numpy()
    a_classes, a_inds = hsn_ref.adp_class_tables()
    valid = a_classes["valid_morph"]
    ref = {}
    for lo, hi in ((0, 2), (2, 3)):
        seeds = np.zeros((hi - lo, len(valid), 41, 41))
        seeds[:, a_inds["morph2valid"]] = H[lo:hi][:, a_inds["all2morph"]]
        bgm = np.stack([scipy.ndimage.gaussian_filter(0.75 * scipy.special.expit(4 * (raw[i].mean(-1) - 240)), sigma=2)
                        for i in range(lo, hi)])
        bgm = torch.nn.functional.interpolate(torch.from_numpy(bgm)[:, None], (41, 41), mode="bilinear",
                                              align_corners=False).numpy()[:, 0]
        seeds[:, 0] = bgm - seeds[:, [valid.index(c) for c in ("A.W", "A.B", "A.M")]].max(1)
        ci = [np.array(a_inds["morph2valid"])[ip[i][:28]] for i in range(lo, hi)]
        cues.update_cues_adp(ref, seeds, ci, list(range(lo, hi)), 0.2)
    for i in range(3):
        assert np.array_equal(out["morph"]["%d_labels" % i], ref["%d_labels" % i])
        a, b = out["morph"]["%d_cues" % i], ref["%d_cues" % i]
        la, lb = np.zeros((41, 41), np.int64), np.zeros((41, 41), np.int64)
        la[a[1], a[2]] = a[0] + 1
        lb[b[1], b[2]] = b[0] + 1
        assert (la == lb).mean() >= 0.99
        assert out["func"]["%d_labels" % i][0] == 1 and out["func"]["%d_cues" % i].shape[0] == 3


def _keras_list_from_state_dict(sd, root, batchnorm, use_bias):
    """model.get_weights() order of the Keras CNN whose transplant is `sd` (inverse of net.common.state_dict_from_keras_weights)."""
    from wsscam.net import common

    out = []
    for kind, key in common.plain_module_order(root, batchnorm):
        if kind == "conv":
            out += [np.transpose(np.asarray(sd[key + ".weight"]), (2, 3, 1, 0)), np.asarray(sd[key + ".bias"])]
        elif kind == "bn":
            out += [np.asarray(sd[key + "." + n]) for n in ("weight", "bias", "running_mean", "running_var")]
        else:
            out.append(np.transpose(np.asarray(sd[key + ".weight"])))
            if use_bias:
                out.append(np.asarray(sd[key + ".bias"]))
    return out


def test_gen_cues_reference_call_form(tmp_path, monkeypatch):
    """gen_cues(dataset, model_type, thresh, batch_size) -- the reference's own call form (02_cues/demo.py:26): settings.ini,
    the session's Keras weight / threshold files under MODEL_ROOT, the split's CSV and images under DATA_ROOT are read through
    wsscam.keras_store, the pickle lands under the cues root.  Same cues as the call with the loaded objects handed in.
    (h5py is not installed here: the .h5 reader is replaced by one that reads the same weight list from an .npz.)"""
    import pickle

    import scipy.io
    from PIL import Image

    from wsscam import keras_store, synth
    from wsscam.cues import demo as cues_demo
    from wsscam.cues import utilities as cues
    from wsscam.net import common, m7_cam

    C = 7
    db = tmp_path / "database"
    dg = db / "DGdevkit"
    os.makedirs(dg / "JPEGImages")
    os.makedirs(dg / "ImageSets" / "Segmentation")
    os.makedirs(db / "models_cnn" / "DeepGlobe_M7")
    ini = tmp_path / "settings.ini"
    ini.write_text("[Download Directory]\ndata_dir = %s\n\n[Data Folders]\nmodel_cnn_dir = models_cnn\ncues_dir = cues\n" % db)
    rng = np.random.default_rng(21)
    names = ["t%d.png" % i for i in range(3)]
    images = [cnn_ref.synth_image(rng, 224, 224) for _ in names]
    labels = (rng.random((3, C)) < 0.4).astype(np.float32)
    labels[:, 1] = 1
    for n, im in zip(names, images):
        Image.fromarray(im).save(str(dg / "JPEGImages" / n))
    header = "Patch Names," + ",".join(keras_store.DEEPGLOBE_CLASSES)
    rows = [n + "," + ",".join(str(int(v)) for v in lb) for n, lb in zip(names, labels)]
    (dg / "ImageSets" / "Segmentation" / "train75.csv").write_text(header + "\n" + "\n".join(rows) + "\n")
    (dg / "ImageSets" / "Segmentation" / "test.csv").write_text(header + "\n" + rows[0] + "\n")
    sd = synth.plain_state_dict("m7", C, True, seed=5)
    thr = np.full((1, C), 0.4)
    mdir = db / "models_cnn" / "DeepGlobe_M7"
    np.savez(str(mdir / "DeepGlobe_M7.npz"), *_keras_list_from_state_dict(sd, "m7", True, True))
    (mdir / "DeepGlobe_M7.h5").write_bytes(b"placeholder")
    scipy.io.savemat(str(mdir / "DeepGlobe_M7.mat"), {"optimalScoreThresh": thr})

    def fake_h5(path):
        z = np.load(path[:-3] + ".npz")
        return [z["arr_%d" % i] for i in range(len(z.files))]

    monkeypatch.setattr(common, "keras_h5_weight_list", fake_h5)
    ds = keras_store.Dataset("DeepGlobe", 224, 2, database_dir=str(db))
    assert ds.sets == ["train75", "test"] and ds.set_gens["train75"].filenames == names
    assert np.array_equal(ds.set_gens["train75"].data, labels)
    out = cues_demo.gen_cues("DeepGlobe", "M7", 0.2, 2, is_verbose=False, settings=str(ini))
    saved = pickle.load(open(db / "cues" / "DeepGlobe_M7" / "localization_cues.pickle", "rb"))
    assert sorted(saved) == sorted(out)
    # the same through the explicit objects
    model = m7_cam.CAM(None, "deepglobe", "M7", C, None)
    model.load_state_dict(dict(sd))
    model.cuda(0)
    alpha = cues.get_grad_cam_weights(model, cues.find_final_layer(model), np.zeros((1, 224, 224, 3)))
    ref = cues_demo.gen_cues("DeepGlobe", "M7", 0.2, 2, is_verbose=False, models={"fg": model}, alphas={"fg": alpha},
                             thresholds={"fg": thr}, images=images, labels=labels, out_dir=str(tmp_path / "explicit"))
    assert sorted(ref) == sorted(out)
    for k in ref:
        assert np.array_equal(np.asarray(ref[k]), np.asarray(out[k])), k
    assert any(np.asarray(out["%d_cues" % i]).size for i in range(3))


@pytest.mark.parametrize("precision", [_lib.PREC_F16X3, _lib.PREC_F16])
def test_cam_head_stream_equals_tiled_head(resnet_sd, precision):
    """The 1x1 CAM head as a streaming GEMM (csrc/cam_head.hip: operands straight from memory into the MFMA, K in four
    quarters added in a fixed order) against the same head through the tiled conv kernel (ctx option OPT_CAM_HEAD_STREAM = 0):
    the same products, a different fp32 summation order -- equal to fp32 round-off; ResNet50 (K = 2048, C = 20, ragged last
    block: 2 x 2 x 5 x 5 rows), VGG16 Grad-CAM with 31 classes, ReLU and scores, and the batch-independence of a row."""
    model = _model(resnet50_cam.CAM, resnet_sd, 20, precision)
    rng = np.random.default_rng(51)
    x = np.stack([cnn_ref.msf_pack(cnn_ref.synth_image(rng, 70, 90), (65, 65)) for _ in range(2)])
    cam = model.forward_batch(x)
    with model.ctx.option(_lib.OPT_CAM_HEAD_STREAM, 0):
        cam_t = model.forward_batch(x)
    tol = 2e-6 if precision == _lib.PREC_F16X3 else 2e-6
    assert cam.shape == cam_t.shape == (2, 20, 5, 5)
    assert np.abs(cam - cam_t).max() <= tol * cam_t.max(), np.abs(cam - cam_t).max() / cam_t.max()
    assert np.array_equal(model.forward_batch(x[1:])[0], cam[1])  # a row's sum does not depend on the rest of the batch
    C = 31
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=8)
    m = _model(vgg16_cam.CAM, sd, C, precision)
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    from wsscam.cues import utilities as cues

    imgs = cnn_ref.normalize_int(np.stack([cnn_ref.synth_image(rng, 65, 65) for _ in range(3)]).astype(np.float64))
    a, sa = cues.conv_and_cams(m, alpha, imgs, relu=True, want_scores=True)
    with m.ctx.option(_lib.OPT_CAM_HEAD_STREAM, 0):
        b, sb = cues.conv_and_cams(m, alpha, imgs, relu=True, want_scores=True)
    assert a.shape == b.shape == (3, 8, 8, C) and (a >= 0).all()
    assert np.abs(a - b).max() <= 2e-6 * b.max() and np.array_equal(sa, sb)


def test_eval_cues_adp_vs_reference_loop(tmp_path):
    """02_cues/demo.py:487-640 (eval_cues_adp): the ADP seeds of both HTT types scored against colour-coded ground truth --
    per class `cv2.resize(cues[:, :, k], (size, size), INTER_NEAREST) == 1` vs the ground-truth colour mask: intersects, unions,
    predicted / gt totals, IoU (no epsilon), the reference's 'precision' / 'recall' ratios and the metrics file.  The device
    counters against the reference's loop restated in numpy on the SAME cues: exact."""
    from tests.test_gpu_edge import _adp_like_image, _cv2_nearest
    from wsscam.cues import demo as cues_demo
    from wsscam.step.eval_cam import ADP_CLS_COLOURS

    C, S = 31, 224
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, C, False, seed=33)
    model = _model(vgg16_cam.CAM, sd, C, _lib.PREC_F16X3)
    rng = np.random.default_rng(34)
    images = [_adp_like_image(rng, S, S) for _ in range(3)]
    alpha = cnn_ref.grad_cam_weights(sd, "vgg16", cnn_ref.VGG16_CFG, 33, C)
    gts = {h: [np.asarray(ADP_CLS_COLOURS[h], np.uint8)[rng.integers(0, len(ADP_CLS_COLOURS[h]), (S, S))] for _ in images]
           for h in ("morph", "func")}
    for h in gts:
        gts[h][0][:5, :7] = (1, 2, 3)  # pixels of no class colour
    out, cues = cues_demo.eval_cues_adp("VGG16", "ADP_tuning_VGG16", 2, S, "tuning", False, False, model=model, alpha=alpha,
                                        thresholds=np.full((1, C), 0.5), images=images, gts=gts, out_dir=str(tmp_path))
    for h in ("morph", "func"):
        cols = ADP_CLS_COLOURS[h]
        n = len(cols)
        inter, union, ptot, gtot = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
        for j in range(3):
            ci = cues[h]["%d_cues" % j]
            cu = np.zeros((41, 41, n))
            cu[ci[1], ci[2], ci[0]] = 1.0
            g = gts[h][j]
            for k, col in enumerate(cols):
                gm = (g[:, :, 0] == col[0]) & (g[:, :, 1] == col[1]) & (g[:, :, 2] == col[2])
                pm = _cv2_nearest(cu[:, :, k], (S, S)) == 1.0
                inter[k] += np.sum(gm & pm)
                union[k] += np.sum(gm | pm)
                ptot[k] += np.sum(pm)
                gtot[k] += np.sum(gm)
        o = out[h]
        assert np.array_equal(o["intersects"], inter) and np.array_equal(o["unions"], union)
        assert np.array_equal(o["predicted_totals"], ptot) and np.array_equal(o["gt_totals"], gtot)
        with np.errstate(divide="ignore", invalid="ignore"):
            assert np.array_equal(o["IoU"], inter / union, equal_nan=True)
        assert np.array_equal(o["Precision"], inter / (gtot + 1e-5)) and np.array_equal(o["Recall"], inter / (ptot + 1e-5))
        assert ptot.sum() > 0  # some seeds exist: the comparison is not vacuous
        rows = open(tmp_path / ("metrics_ADP-%s_tuning_VGG16.csv" % h)).read().strip().splitlines()
        assert rows[0] == ",Class,IoU,Precision,Recall" and len(rows) == n + 2
