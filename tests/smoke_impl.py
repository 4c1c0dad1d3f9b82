"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.

ResNet50-CAM (2 images, 97x97 network input) -> make_cam tail -> [bg | maps] unaries -> dense-CRF,
all through the C ABI; compared with the torch fp32 restatement and the C CRF restatement."""
import numpy as np
import torch

from oracle import cnn_ref
from tests import helpers
from wsscam import _lib
from wsscam.net import resnet50_cam
from wsscam.step import make_cam


def run():
    S, C = 97, 20
    sd = cnn_ref.make_resnet50_cam_state_dict(C, seed=0)
    rng = np.random.default_rng(1)
    sizes = [(120, 150), (97, 97)]
    imgs = [cnn_ref.synth_image(rng, *sz) for sz in sizes]
    labels = [np.zeros(C, np.float32), np.zeros(C, np.float32)]
    labels[0][[2, 9]] = 1
    labels[1][[14]] = 1
    packs = [{"name": "smoke%d" % i, "img": cnn_ref.msf_pack(im, (S, S)), "size": sz, "label": lb}
             for i, (im, sz, lb) in enumerate(zip(imgs, sizes, labels))]

    class Args:
        split = "train_aug"
        dataset = "voc12"
        cam_out_dir = None

    model = resnet50_cam.CAM(None, "voc12", "", C, None, precision=_lib.PREC_F16X3)  # the headline (fp32-class) mode
    model.load_state_dict(sd)
    model.eval().cuda(0)
    arch, cus = model.ctx.device_info()
    outs = make_cam.process_batch(model, packs, Args, save=False)
    worst = 0.0
    for p, o in zip(packs, outs):
        ref = cnn_ref.make_cam_image(torch.from_numpy(p["img"]), sd, p["size"], torch.from_numpy(p["label"]))
        assert np.array_equal(o["keys"], ref["keys"])
        worst = max(worst, float(np.abs(o["high_res"] - ref["high_res"]).max()),
                    float(np.abs(o["cam"] - ref["cam"]).max()))
    assert worst <= 1e-4, "CAM parity %.3g > 1e-4" % worst

    # CRF: a non-saturated case (soft class probabilities that follow, but do not equal, the image regions; M = 5) against
    # the C oracle -- round 3's smoke fed both sides the near-one-hot [bg | GT map] unaries and landed at 1e-14
    ctx = model.ctx
    rgb, U, _ = helpers.synth_crf_case(np.random.default_rng(5), 97, 97, 5, sharp=2.0)
    cfg = (1.5, 3, 40, 13, 10, 10)
    # (a size's second use brings the tile vertex sets of the on-chip Gaussian message: the object the check runs on takes the
    # production path -- the message formed inside the update kernel)
    _lib.Crf(ctx, ctx.to_device(rgb), 1, 97, 97, cfg[0], cfg[2], cfg[3]).close()
    crf = _lib.Crf(ctx, ctx.to_device(rgb), 1, 97, 97, cfg[0], cfg[2], cfg[3])
    assert crf.gaussian_on_chip(5)
    q_dev = ctx.alloc(5 * 97 * 97 * 4)
    a_dev = ctx.alloc(97 * 97 * 4)
    crf.inference(ctx.to_device(U), 5, cfg[1], cfg[4], cfg[5], q_dev, a_dev)
    q = ctx.to_host(q_dev, (5, 97 * 97), np.float32)
    a = ctx.to_host(a_dev, (97 * 97,), np.int32)
    crf.close()
    qr, ar, _ = helpers.crf_oracle(rgb, U, cfg)
    dq = float(np.abs(q - qr).max())
    agree = float((a == ar).mean())
    assert dq <= 1e-3 and agree >= 0.995, (dq, agree)
    assert len(np.unique(ar)) >= 3 and float(np.abs(qr - 0.5).min()) < 0.45, "degenerate smoke case"
    # chain level: CNN -> unaries -> CRF labels of the first image at 97 x 97 against the all-fp32 oracle chain
    x0 = packs[1]["img"][None]
    lab = helpers.product_chain(ctx, model._ensure_net(), x0, imgs[1][None], cfg, C)
    ref_lab, _, _ = helpers.oracle_chain(packs[1]["img"], imgs[1], {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, cfg, C)
    chain = float((lab[0] == ref_lab).mean())
    assert chain >= 0.995, chain
    print("smoke ok on %s (%d CUs): CAM max|d| %.2e (f16x3), CRF max|dQ| %.2e, label agreement %.4f, chain label agreement %.5f"
          % (arch, cus, worst, dq, agree, chain))
