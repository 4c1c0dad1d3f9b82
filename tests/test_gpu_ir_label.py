"""GPU: cam_to_ir_label.run (03b_irn/step/cam_to_ir_label.py:18-117) for the three dataset branches -- VOC (two CRF
runs, fg/bg merge :42-58), ADP (one run, keys = [-1] + keys :27-40) and DeepGlobe (image /4, `cam` maps :60-73) -- through
the step driver: same-size batching, image sharding over two workers (both on GPU 0), label / colour / overlay PNGs.
Every label map against the oracle chain: numpy arg-max + unary_from_labels + the C dense-CRF restatement."""
import argparse
import os

import numpy as np
import pytest
from PIL import Image

from tests import helpers
from wsscam.misc import imutils
from wsscam.step import cam_to_ir_label

pytestmark = pytest.mark.gpu

CFG = (3, 3, 50, 5, 10, 10)


def _oracle_pred(rgb, maps, thres):
    lab = np.argmax(np.pad(maps, ((1, 0), (0, 0), (0, 0)), mode="constant", constant_values=thres), axis=0)
    U = imutils.unary_from_labels(lab, maps.shape[0] + 1, 0.7, zero_unsure=False)
    _, ar, _ = helpers.crf_oracle(rgb, U, CFG)
    return ar.reshape(rgb.shape[:2])


@pytest.mark.parametrize("dataset", ["voc12", "adp_morph", "deepglobe"])
def test_cam_to_ir_label_run(tmp_path, dataset):
    rng = np.random.default_rng({"voc12": 1, "adp_morph": 2, "deepglobe": 3}[dataset])
    cam_dir, out_dir, clr_dir = str(tmp_path / "cam"), str(tmp_path / "ir"), str(tmp_path / "clr")
    os.makedirs(cam_dir)
    data, expect = [], {}
    sizes = [(47, 59), (47, 59), (59, 47), (47, 59), (40, 40)] if dataset != "deepglobe" else [(96, 96)] * 3 + [(64, 64)]
    for i, (H, W) in enumerate(sizes):
        rgb, _, p = helpers.synth_crf_case(rng, H, W, 3)
        K = 2 if i != 3 else 1
        keys = np.array([4, 11][:K]) if dataset != "deepglobe" else np.array([1, 5][:K])
        name = "img%02d" % i
        if dataset == "deepglobe":
            h4 = H // 4
            _, _, p4 = helpers.synth_crf_case(rng, h4, h4, 3)
            maps = (p4[:K] / p4[:K].max(axis=(1, 2), keepdims=True)).astype(np.float32)
            np.save(os.path.join(cam_dir, name + ".npy"), {"keys": keys, "cam": maps})
            small = cam_to_ir_label._nearest_resize_cv2(rgb, (H // 4, W // 4))
            k = np.concatenate((np.array([-1]), keys))
            conf = k[_oracle_pred(small, maps, 0.3)]
            conf[conf == -1] = 255
        else:
            maps = (p[:K] / p[:K].max(axis=(1, 2), keepdims=True)).astype(np.float32)
            if i == 4:
                keys, maps = np.empty(0), np.empty(0)
            np.save(os.path.join(cam_dir, name + ".npy"), {"keys": keys, "cam": np.empty(0), "high_res": maps})
            if i == 4:
                conf = np.full((H, W), 0 if dataset == "voc12" else 255)
            elif dataset == "voc12":
                k = np.pad(keys + 1, (1, 0), mode="constant")
                fg, bg = k[_oracle_pred(rgb, maps, 0.30)], k[_oracle_pred(rgb, maps, 0.05)]
                conf = fg.copy()
                conf[fg == 0] = 255
                conf[bg + fg == 0] = 0
            else:
                k = np.concatenate((np.array([-1]), keys))
                conf = k[_oracle_pred(rgb, maps, 0.30)]
                conf[conf == -1] = 255
        data.append({"name": name, "img": rgb})
        expect[name] = conf.astype(np.uint8)
    colours = {"bg": [(0, 0, 0)], "fg": [(10 * i + 5, 200 - 7 * i, 30 + 4 * i) for i in range(20)]}
    args = argparse.Namespace(dataset=dataset, cam_out_dir=cam_dir, ir_label_out_dir=out_dir, ir_label_clr_out_dir=clr_dir,
                              conf_fg_thres=0.30, conf_bg_thres=0.05, dataset_obj=data, num_workers=2, cam_device_ids=[0, 0],
                              class_colours=colours, overlay_r=0.75, ir_label_batch_images=2)
    cam_to_ir_label.run(args)
    assert sorted(os.listdir(out_dir)) == sorted(n + ".png" for n in expect)
    for name, ref in expect.items():
        got = np.asarray(Image.open(os.path.join(out_dir, name + ".png")))
        assert got.shape == ref.shape and got.dtype == np.uint8
        assert (got == ref).mean() >= 0.995, (name, (got == ref).mean())
        assert set(np.unique(got)) <= set(np.unique(ref)) | {0, 255}
        clr = np.asarray(Image.open(os.path.join(clr_dir, name + ".png")))
        assert clr.shape == ref.shape + (3,) and np.all(clr[got == 255] == 255)
        assert os.path.exists(os.path.join(clr_dir, name + "_overlay.png"))


@pytest.mark.parametrize("mode", ["voc12", "fg"])
def test_ir_label_ragged_equals_per_group(ctx, mode):
    """A mixed-size, mixed-class-count list through ONE ragged CRF object (what cam_to_ir_label._work now issues per device
    batch) gives, bit for bit, the label maps of the per-(H, W, K) group path (ir_label_batch), which is the reference's
    per-image loop (cam_to_ir_label.py:25-58) batched."""
    rng = np.random.default_rng(99)
    specs = [(47, 59, 2), (59, 47, 1), (47, 59, 3), (47, 59, 2), (40, 40, 4), (59, 47, 1), (47, 59, 2)]
    items = []
    for (H, W, K) in specs:
        rgb, _, p = helpers.synth_crf_case(rng, H, W, K + 1)
        maps = (p[:K] / p[:K].max(axis=(1, 2), keepdims=True)).astype(np.float32)
        items.append((rgb, maps, np.sort(rng.choice(20, K, replace=False))))
    got = cam_to_ir_label.ir_label_ragged(ctx, items, mode, 0.30, 0.05)
    groups = {}
    for i, s in enumerate(specs):
        groups.setdefault(s, []).append(i)
    for s, idx in groups.items():
        ref = cam_to_ir_label.ir_label_batch(ctx, np.stack([items[i][0] for i in idx]), np.stack([items[i][1] for i in idx]),
                                             [items[i][2] for i in idx], mode, 0.30, 0.05)
        for j, i in enumerate(idx):
            assert got[i].shape == ref[j].shape and np.array_equal(got[i], ref[j]), (s, i)
