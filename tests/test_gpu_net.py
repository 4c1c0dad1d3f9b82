"""GPU: whole CAM networks + CAM tail through the C ABI against the torch fp32 oracle.

Stated tolerances on the per-class max-normalised CAM maps make_cam writes (values in [0,1]),
max over every pixel of every class of every image, vs the fp32 oracle:
  f16    (default product precision): max|d| <= 2e-2      (measured 1.5e-2, mean 3e-4)
  bf16   (fast mode)                : max|d| <= 1e-1      (measured 8.3e-2, mean 2e-3)
  bf16x3 (fp32-class mode)          : max|d| <= 2e-4      (measured 1.2e-4; the torch-CPU oracle itself
                                                           moves by 3e-5 with its reduction order)
and on the raw (un-normalised) CAM: f16 5e-3, bf16 3e-2, bf16x3 2e-4 (x max(cam)).
"""
import numpy as np
import pytest
import torch

from oracle import cnn_ref
from wsscam import _lib
from wsscam.net import resnet50_cam, vgg16_cam, m7_cam

pytestmark = pytest.mark.gpu

TOL_NORM = {_lib.PREC_BF16: 1e-1, _lib.PREC_F16: 2e-2, _lib.PREC_BF16X3: 2e-4}
TOL_RAW = {_lib.PREC_BF16: 3e-2, _lib.PREC_F16: 5e-3, _lib.PREC_BF16X3: 2e-4}
PRECISIONS = [_lib.PREC_BF16, _lib.PREC_F16, _lib.PREC_BF16X3]


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
            assert agree >= {_lib.PREC_BF16: 0.98, _lib.PREC_F16: 0.995, _lib.PREC_BF16X3: 0.9999}[precision], agree


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
                                                       _lib.PREC_BF16X3: 1e-4}[precision]


def test_m7_cam():
    C = 20
    sd = cnn_ref.make_plain_state_dict("m7", cnn_ref.M7_CFG, C, True, seed=2)
    alpha = cnn_ref.grad_cam_weights(sd, "m7", cnn_ref.M7_CFG, 32, C)  # (F, C), 02_cues/utilities.py:60-99
    sd_dev = dict(sd)
    sd_dev["gradcam_weights"] = torch.from_numpy(alpha.astype(np.float32))
    model = _model(m7_cam.CAM, sd_dev, C, _lib.PREC_BF16X3)
    rng = np.random.default_rng(5)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 70, 60), (64, 64))
    cam, score = model.forward_batch(x[None], want_score=True)
    with torch.no_grad():
        rcam, rscore = cnn_ref.m7_cam_forward(torch.from_numpy(x), sd, torch.from_numpy(alpha), C)
    assert cam[0].shape == tuple(rcam.shape) == (C, 16, 16)
    assert np.abs(cam[0] - rcam.numpy()).max() <= 2e-4 * max(float(rcam.max()), 1e-3)
    assert np.abs(score[0] - rscore.numpy()).max() <= 1e-4


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
