"""GPU: the MSF dataset transform on the device (wsc_msf_input_u8, csrc/input.hip) is BIT-IDENTICAL to the ORACLE's
statement of 03b_irn/voc12/dataloader.py:68-106, 225-246 (oracle/cnn_ref.py::msf_pack / resize_bilinear_f64 /
normalize_*: float64 bilinear resize, float32 normalisation, CHW, flip pair) -- not to the product's own numpy mirror,
which tests/test_dataloaders_host.py holds to the same oracle on the CPU -- and make_cam.run fed decoded uint8 images
writes the same files as the run fed the float32 pairs."""
import argparse
import os

import numpy as np
import pytest

from oracle import cnn_ref
from wsscam import _lib, synth
from wsscam.adp import dataloader as adp_dl
from wsscam.step import make_cam
from wsscam.voc12 import dataloader as voc_dl

pytestmark = pytest.mark.gpu


def _pack_u8(ctx, imgs):
    offs = np.concatenate(([0], np.cumsum([im.size for im in imgs]))).astype(np.int64)
    flat = np.concatenate([im.reshape(-1) for im in imgs])
    return ctx.to_device(flat), offs[:-1], [im.shape[:2] for im in imgs]


@pytest.mark.parametrize("mode", ["int", "float"])
def test_msf_input_bit_identical(ctx, mode):
    rng = np.random.default_rng(5)
    for S, shapes in ((321, [(375, 500), (500, 333), (321, 321), (97, 640)]), (224, [(240, 200), (224, 224), (1, 7)])):
        imgs = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for h, w in shapes]
        norm = voc_dl.TorchvisionNormalize(mode)
        ref = np.stack([cnn_ref.msf_pack(im, (S, S), mode) for im in imgs])  # the oracle, not the product's mirror
        dev, offs, sizes = _pack_u8(ctx, imgs)
        x_dev = ctx.alloc(ref.nbytes)
        _lib.msf_input_u8(ctx, dev, sizes, offs, S, norm.mean, norm.std, x_dev, pre_div255=mode == "float", pair=True)
        got = ctx.to_host(x_dev, ref.shape, np.float32)
        assert np.array_equal(got, ref), (S, mode, np.abs(got - ref).max())
    # ADP constants, plain batch (02_cues / 03c_hsn read_batch + normalise; no flip pair)
    imgs = [rng.integers(100, 256, (272, 272, 3)).astype(np.uint8) for _ in range(2)]
    an = adp_dl.TorchvisionNormalize("int")
    def adp_norm(x):  # 03c_hsn/utilities.py / adp dataloader: (x - mean) / std with the ADP constants, float32 like normalize_int
        out = np.empty(x.shape, np.float32)
        xf = np.float32(x)
        for c in range(3):
            out[..., c] = (xf[..., c] - an.mean[c]) / an.std[c]
        return out

    ref = np.stack([np.transpose(adp_norm(cnn_ref.resize_bilinear_f64(im, (224, 224))), (2, 0, 1)) for im in imgs]).astype(np.float32)
    dev, offs, sizes = _pack_u8(ctx, imgs)
    x_dev = ctx.alloc(ref.nbytes)
    _lib.msf_input_u8(ctx, dev, sizes, offs, 224, an.mean, an.std, x_dev, pre_div255=False, pair=False)
    assert np.array_equal(ctx.to_host(x_dev, ref.shape, np.float32), ref)


def test_resize_u8_device_bit_identical(ctx):
    """wsc_resize_u8 (cv2.resize on uint8, fixed point: read_batch of 02_cues / 03c_hsn) == the numpy statement == the
    oracle's loop statement, bit for bit, on VOC-like, ADP-like and degenerate sizes incl. the 2 x 2 decimation shortcut."""
    from oracle import hsn_ref
    from wsscam.cues import demo as cues_demo

    rng = np.random.default_rng(11)
    for (oh, ow), shapes in (((321, 321), [(375, 500), (500, 333), (321, 321), (97, 640), (642, 642)]),
                             ((224, 224), [(272, 272), (224, 224), (1, 7), (448, 448)]), ((40, 56), [(33, 90), (80, 112)])):
        imgs = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for h, w in shapes]
        dev, offs, sizes = _pack_u8(ctx, imgs)
        out_dev = ctx.alloc(len(imgs) * oh * ow * 3)
        _lib.resize_u8(ctx, dev, sizes, offs, (oh, ow), out_dev)
        got = ctx.to_host(out_dev, (len(imgs), oh, ow, 3), np.uint8)
        ref = np.stack([voc_dl.resize_bilinear_u8(im, (oh, ow)) for im in imgs])
        assert np.array_equal(got, ref), ((oh, ow), np.abs(got.astype(int) - ref.astype(int)).max())
        assert np.array_equal(cues_demo.read_batch_u8(imgs, (oh, ow), ctx=ctx), ref)
    small = [rng.integers(0, 256, (23, 31, 3)).astype(np.uint8), rng.integers(0, 256, (40, 20, 3)).astype(np.uint8)]
    assert np.array_equal(cues_demo.read_batch_u8(small, (33, 33), ctx=ctx), hsn_ref.read_batch_u8(small, (33, 33)))


def test_make_cam_device_transform_same_files(tmp_path):
    rng = np.random.default_rng(6)
    sd = synth.resnet50_cam_state_dict(20, seed=2)
    norm = voc_dl.TorchvisionNormalize("int")
    S = 129
    a, b = [], []
    for i in range(7):
        H0, W0 = [(60, 80), (80, 60), (129, 129)][i % 3]
        img = synth.synth_image(rng, H0, W0)
        lab = np.zeros(20, np.float32)
        lab[[i % 20, (7 * i + 3) % 20]] = 1
        a.append({"name": "im%02d" % i, "img": voc_dl.msf_pack(img, (S, S), norm), "size": (H0, W0), "label": lab})
        b.append({"name": "im%02d" % i, "img_u8": img, "size": (H0, W0), "label": lab})

    def args(out, packs):
        return argparse.Namespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                                  use_cls=list(range(20)), model_id="resnet50", state_dict=sd, split="train_aug", dataset_obj=packs,
                                  cam_out_dir=out, outsize=(S, S), n_gpus=1, cam_batch_images=4, cam_precision=_lib.PREC_F16,
                                  cam_weights_name="unused", norm_mode="int", val_list=None, dev_root=None, cam_scales=(1.0,),
                                  class_names={"bg": ["background"], "fg": ["c%d" % i for i in range(20)]})

    d1, d2, d3 = str(tmp_path / "host"), str(tmp_path / "dev"), str(tmp_path / "serial")
    make_cam.run(args(d1, a))
    make_cam.run(args(d2, b))
    # the SERIAL path (cam_pipeline off: the documented one-batch-at-a-time path, also what a multi-GPU run with the pipeline
    # disabled takes) must accept the decoded-image items too (it raised KeyError 'img'): same files again
    a3 = args(d3, b)
    a3.cam_pipeline = False
    make_cam.run(a3)
    for f in sorted(os.listdir(d1)):
        x = np.load(os.path.join(d1, f), allow_pickle=True).item()
        y = np.load(os.path.join(d2, f), allow_pickle=True).item()
        z = np.load(os.path.join(d3, f), allow_pickle=True).item()
        for k in x:
            assert np.array_equal(x[k], y[k]), (f, k)
            assert np.array_equal(x[k], z[k]), (f, k, "serial path")
    # the default of the device-side transform follows the worker count run() will use and the path that will consume the items
    ns = argparse.Namespace(n_gpus=2, outsize=(S, S), dataset="voc12")
    assert make_cam._device_transform_default(ns) is True
    ns.cam_pipeline = False
    assert make_cam._device_transform_default(ns) is False
    assert make_cam._device_transform_default(argparse.Namespace(n_gpus=2, outsize=None)) is False
    assert make_cam._device_transform_default(argparse.Namespace(n_gpus=1, outsize=(S, S))) is False
    assert make_cam._device_transform_default(argparse.Namespace(n_gpus=1, outsize=(S, S), cam_device_transform=True)) is True
