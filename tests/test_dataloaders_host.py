"""CPU: the three MSF dataset mirrors (03b_irn/{voc12,adp,deepglobe}/dataloader.py) on synthetic image files --
item layout, per-dataset normalisation constants, flip pair, label lookup, name-list parsing."""
import os

import numpy as np
import pytest

from wsscam.adp import dataloader as adp_dl
from wsscam.deepglobe import dataloader as dg_dl
from wsscam.voc12 import dataloader as voc_dl

PIL_Image = pytest.importorskip("PIL.Image")


def _write(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    PIL_Image.fromarray(arr).save(path)


def test_adp_dataset_item(tmp_path):
    rng = np.random.default_rng(0)
    names = ["001.png_crop_1", "002.png_crop_7"]
    imgs = [rng.integers(120, 256, (240, 200, 3)).astype(np.uint8), rng.integers(0, 256, (224, 224, 3)).astype(np.uint8)]
    for n, im in zip(names, imgs):
        _write(str(tmp_path / "PNGImagesSubset" / (n + ".png")), im)
    lst = tmp_path / "evaluation.txt"
    lst.write_text("% comment line\n" + "\n".join(names) + "\n")
    labels = {names[0]: np.array([1, 0, 1]), names[1]: np.array([0, 1, 0])}
    lp = tmp_path / "cls_labels_func.npy"
    np.save(lp, labels)
    ds = adp_dl.ADPClassificationDatasetMSF(str(lst), str(tmp_path), "func", True, norm_mode="int", outsize=(224, 224),
                                            cls_labels_path=str(lp))
    assert len(ds) == 2
    it = ds[1]  # already 224x224: no resize, exact normalisation check
    assert it["name"] == names[1] and it["size"] == (224, 224)
    assert it["img"].shape == (2, 3, 224, 224) and it["img"].dtype == np.float32
    ref = (np.float32(imgs[1]) - np.float32(193.09203)) / np.float32(56.450138)
    assert np.allclose(it["img"][0], ref.transpose(2, 0, 1), atol=1e-6)
    assert np.array_equal(it["img"][1], it["img"][0][:, :, ::-1])
    assert it["orig_img"].dtype == np.uint8 and np.array_equal(it["orig_img"][0], imgs[1])
    # reference quirk kept: np.flip(s_img_orig, -1) on an HWC array reverses the CHANNELS, not the width
    # (adp/dataloader.py:232); only orig_img[0] is consumed (common_cam.py:36)
    assert np.array_equal(it["orig_img"][1], imgs[1][:, :, ::-1])
    assert np.array_equal(it["label"], labels[names[1]])
    it0 = ds[0]
    assert it0["img"].shape == (2, 3, 224, 224) and it0["orig_img"].shape == (2, 240, 200, 3) and it0["size"] == (240, 200)
    f = adp_dl.TorchvisionNormalize("float")(np.full((2, 2, 3), 255.0))
    assert np.allclose(f, (1.0 - 0.757) / 0.221)


def test_deepglobe_dataset_item(tmp_path):
    rng = np.random.default_rng(1)
    names = ["208695", "334677"]
    im = rng.integers(0, 256, (300, 260, 3)).astype(np.uint8)
    for n in names:
        _write(str(tmp_path / "JPEGImages" / (n + ".jpg")), im)
    lst = tmp_path / "test.txt"
    lst.write_text("\n".join(names) + "\n")
    # label table in the reference's format: {name: 7-vector incl. 'unknown'} (deepglobe/dataloader.py:29-35)
    lp = tmp_path / "cls_labels_balanced.npy"
    np.save(str(lp), {names[0]: np.array([1, 0, 0, 1, 0, 0, 0.]), names[1]: np.array([0, 1, 0, 0, 0, 0, 1.])})
    ds = dg_dl.DeepGlobeClassificationDatasetMSF(str(lst), str(tmp_path), True, norm_mode="int", outsize=(224, 224),
                                                 cls_labels_path=str(lp))
    it = ds[0]
    assert it["img"].shape == (2, 3, 224, 224) and it["orig_img"].shape == (2, 300, 260, 3)
    assert it["label"].shape == (6,)  # 'unknown' dropped (deepglobe/dataloader.py:35)
    assert 0.0 <= it["img"].min() and it["img"].max() <= 1.0  # x / 255
    x = np.float64(np.asarray(PIL_Image.open(dg_dl.get_img_path(names[0], str(tmp_path))).convert("RGB")))
    ref = np.float32(voc_dl.resize_bilinear_f64(x, (224, 224))) / np.float32(255.0)
    assert np.allclose(it["img"][0], ref.transpose(2, 0, 1), atol=1e-6)


def test_voc_dataset_item(tmp_path, monkeypatch):
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, (50, 70, 3)).astype(np.uint8)
    _write(str(tmp_path / "JPEGImages" / "2007_000032.jpg"), im)
    lst = tmp_path / "val.txt"
    lst.write_text("2007_000032\n")
    # label table in the reference's format: {int-coded name: 20-vector} (voc12/dataloader.py:60-66)
    (tmp_path / "voc12").mkdir()
    row = np.zeros(20, np.float32)
    row[[0, 14]] = 1
    np.save(str(tmp_path / "voc12" / "cls_labels.npy"), {int(voc_dl.load_img_name_list(str(lst))[0]): row})
    monkeypatch.setenv("WSSCAM_CLS_LABELS_ROOT", str(tmp_path))  # the lookup order of adp.dataloader.find_cls_labels
    ds = voc_dl.VOC12ClassificationDatasetMSF(str(lst), str(tmp_path), norm_mode="int", outsize=(224, 224))
    it = ds[0]
    assert it["name"] == "2007_000032" and it["size"] == (50, 70)
    assert it["img"].shape == (2, 3, 224, 224)
    assert np.array_equal(it["label"], row)
    # Caffe BGR means applied to R,G,B in that order
    x = np.float64(np.asarray(PIL_Image.open(voc_dl.get_img_path("2007_000032", str(tmp_path))).convert("RGB")))
    r = voc_dl.resize_bilinear_f64(x, (224, 224))
    assert np.allclose(it["img"][0, 0], (np.float32(r[..., 0]) - 104.0) / 255.0, atol=1e-6)
    assert np.allclose(it["img"][0, 2], (np.float32(r[..., 2]) - 123.0) / 255.0, atol=1e-6)
    # args.cam_scales with several entries (voc12/dataloader.py:231-242): a LIST of pairs, every scale PIL-bicubic rescaled to
    # round(H s) x round(W s) and then resized to the same outsize; scale 1 is the plain item; device_transform hands over
    # the rescaled uint8 images instead
    from wsscam.misc import imutils

    ms = voc_dl.VOC12ClassificationDatasetMSF(str(lst), str(tmp_path), norm_mode="int", outsize=(224, 224), scales=(1.0, 0.5, 1.5))[0]
    assert isinstance(ms["img"], list) and [a.shape for a in ms["img"]] == [(2, 3, 224, 224)] * 3
    assert np.array_equal(ms["img"][0], it["img"]) and ms["size"] == (50, 70)
    raw = np.asarray(PIL_Image.open(voc_dl.get_img_path("2007_000032", str(tmp_path))).convert("RGB"))
    half = imutils.pil_rescale(raw, 0.5, order=3)
    assert half.shape == (25, 35, 3) and imutils.pil_rescale(raw, 1.5, order=3).shape == (75, 105, 3)
    assert np.array_equal(half, np.asarray(PIL_Image.fromarray(raw).resize((35, 25), PIL_Image.BICUBIC)))
    assert np.array_equal(ms["img"][1], voc_dl.msf_pack(half, (224, 224), voc_dl.TorchvisionNormalize("int")))
    assert imutils.pil_resize(raw, (50, 70), 3) is raw
    u8 = voc_dl.VOC12ClassificationDatasetMSF(str(lst), str(tmp_path), norm_mode="int", outsize=(224, 224), scales=(1.0, 0.5),
                                              device_transform=True)[0]
    assert "img" not in u8 and [a.shape for a in u8["img_u8"]] == [(50, 70, 3), (25, 35, 3)]


def test_label_table_lookup_order(tmp_path, monkeypatch):
    """adp.dataloader.find_cls_labels: explicit path > $WSSCAM_CLS_LABELS_ROOT > working directory; a clear error when
    the table is nowhere (the package ships no copies of the reference's annotation files)."""
    from wsscam.adp.dataloader import find_cls_labels

    monkeypatch.delenv("WSSCAM_CLS_LABELS_ROOT", raising=False)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError) as ei:
        find_cls_labels(os.path.join("voc12", "no_such_table.npy"))
    assert "WSSCAM_CLS_LABELS_ROOT" in str(ei.value)
    (tmp_path / "voc12").mkdir()
    (tmp_path / "voc12" / "t.npy").write_bytes(b"x")
    assert find_cls_labels(os.path.join("voc12", "t.npy")) == os.path.join("voc12", "t.npy")  # cwd, as the reference
    root = tmp_path / "root"
    (root / "voc12").mkdir(parents=True)
    (root / "voc12" / "t.npy").write_bytes(b"y")
    monkeypatch.setenv("WSSCAM_CLS_LABELS_ROOT", str(root))
    assert find_cls_labels(os.path.join("voc12", "t.npy")) == str(root / "voc12" / "t.npy")
    assert find_cls_labels(os.path.join("voc12", "t.npy"), "/explicit/path.npy") == "/explicit/path.npy"


def _cv2_inter_linear_ref(img, out_hw):
    """INDEPENDENT statement of cv2.resize(..., interpolation=INTER_LINEAR) on float images, scalar loops, straight from
    OpenCV's documented coordinate rule (imgproc resize, `fx = (dx + 0.5) * scale_x - 0.5; sx = floor(fx); fx -= sx;
    sx < 0 -> (sx, fx) = (0, 0); sx >= W - 1 -> (sx, fx) = (W - 1, 0)`; same for y): out = sum of the four taps."""
    H, W = img.shape[:2]
    oh, ow = out_hw
    out = np.zeros((oh, ow) + img.shape[2:], np.float64)
    sy_, sx_ = H / oh, W / ow
    for dy in range(oh):
        fy = (dy + 0.5) * sy_ - 0.5
        y0 = int(np.floor(fy))
        fy -= y0
        if y0 < 0:
            y0, fy = 0, 0.0
        if y0 >= H - 1:
            y0, fy = H - 1, 0.0
        y1 = min(y0 + 1, H - 1)
        for dx in range(ow):
            fx = (dx + 0.5) * sx_ - 0.5
            x0 = int(np.floor(fx))
            fx -= x0
            if x0 < 0:
                x0, fx = 0, 0.0
            if x0 >= W - 1:
                x0, fx = W - 1, 0.0
            x1 = min(x0 + 1, W - 1)
            out[dy, dx] = ((1 - fy) * (1 - fx) * img[y0, x0] + (1 - fy) * fx * img[y0, x1] +
                           fy * (1 - fx) * img[y1, x0] + fy * fx * img[y1, x1])
    return out


def test_resize_bilinear_f64_against_cv2_rule():
    """TorchvisionResize (03b_irn/voc12/dataloader.py:68-78: float64 cv2.resize, bilinear) -- cv2 itself is absent
    offline, so the vectorised resize of the dataloaders is checked against an independent scalar statement of OpenCV's
    INTER_LINEAR rule (up- and down-scaling, non-square, 1-pixel sides), plus closed forms: a linear ramp stays the
    same ramp away from the clamped border, an impulse spreads with the documented hat weights, constants are kept."""
    rng = np.random.default_rng(8)
    for (H, W), out in (((7, 5), (11, 13)), ((20, 31), (9, 8)), ((1, 6), (4, 4)), ((6, 1), (3, 7)), ((12, 12), (12, 12)),
                        ((37, 50), (64, 64))):
        img = rng.normal(size=(H, W, 3)) * 50 + 100
        got = voc_dl.resize_bilinear_f64(img, out)
        ref = _cv2_inter_linear_ref(img, out)
        assert got.shape == ref.shape and got.dtype == np.float64
        assert np.abs(got - ref).max() <= 1e-10, ((H, W), out, np.abs(got - ref).max())
    # ramp: f(x) = a x + b sampled at the half-pixel source coordinate, exact where no clamping happens
    W, ow = 40, 100
    ramp = (3.0 * np.arange(W) + 2.0)[None, :, None].repeat(4, 0)
    r = voc_dl.resize_bilinear_f64(ramp, (4, ow))[0, :, 0]
    src = (np.arange(ow) + 0.5) * W / ow - 0.5
    inner = (src >= 0) & (src <= W - 1)
    assert np.abs(r[inner] - (3.0 * src[inner] + 2.0)).max() <= 1e-10
    assert r[0] == ramp[0, 0, 0] and r[-1] == ramp[0, -1, 0]  # clamped ends
    # impulse at column 5 of 16, upscaled x2: hat weights max(0, 1 - |src - 5|)
    imp = np.zeros((1, 16, 1))
    imp[0, 5, 0] = 1.0
    r = voc_dl.resize_bilinear_f64(imp, (1, 32))[0, :, 0]
    src = (np.arange(32) + 0.5) / 2 - 0.5
    assert np.abs(r - np.maximum(0, 1 - np.abs(src - 5))).max() <= 1e-12
    assert np.all(voc_dl.resize_bilinear_f64(np.full((9, 7, 3), 4.25), (15, 3)) == 4.25)


def test_msf_transform_mirror_equals_oracle_bit_for_bit():
    """The product's host transform (wsscam.voc12.dataloader.msf_pack / resize_bilinear_f64 / TorchvisionNormalize) and
    the oracle's statement of 03b_irn/voc12/dataloader.py:68-106, 225-246 (oracle/cnn_ref.py) are two numpy versions of
    the same lines: they must agree BIT FOR BIT, so that a regression in the shared mirror cannot hide behind the device
    test (which compares the HIP kernel with the oracle) -- VERDICT r5 weak #2."""
    from oracle import cnn_ref

    rng = np.random.default_rng(17)
    for S, shapes in ((321, [(375, 500), (500, 333), (321, 321), (97, 640)]), (224, [(240, 200), (224, 224), (1, 7)])):
        for h, w in shapes:
            im = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            assert np.array_equal(voc_dl.resize_bilinear_f64(im, (S, S)), cnn_ref.resize_bilinear_f64(im, (S, S)))
            for mode in ("int", "float"):
                got = voc_dl.msf_pack(im, (S, S), voc_dl.TorchvisionNormalize(mode))
                ref = cnn_ref.msf_pack(im, (S, S), mode)
                assert got.dtype == ref.dtype == np.float32 and np.array_equal(got, ref), (S, (h, w), mode)
    im = rng.integers(0, 256, (33, 47, 3)).astype(np.uint8)  # outsize None: the image keeps its size
    assert np.array_equal(voc_dl.msf_pack(im, None, voc_dl.TorchvisionNormalize("int")), cnn_ref.msf_pack(im, None))


def test_resize_bilinear_u8_cv2_fixed_point_rule():
    """read_batch of 02_cues/utilities.py:172-176, 03c_hsn/utilities.py:170-181 and the ADP twins keep the resized batch as
    uint8: cv2.resize's 8-bit INTER_LINEAR result (11-bit fixed-point coefficients, two passes, truncating shifts).  cv2 is
    absent offline, so the vectorised host version is checked bit for bit against the oracle's loop statement of OpenCV's
    published algorithm (oracle/hsn_ref.py::cv2_resize_u8), plus properties the fixed-point rule must have: constants are
    kept, a same-size call copies, an exact 2 x 2 decimation is the rounded box mean, and the result stays within one
    level of the float64 bilinear value (the rule TorchvisionResize's float path follows) -- but is NOT its rounding or
    truncation everywhere, which is why the drivers must not quantise the float path."""
    from oracle import hsn_ref

    rng = np.random.default_rng(9)
    n_diff_round = n_diff_trunc = 0
    for (H, W), out in (((7, 5), (11, 13)), ((20, 31), (9, 8)), ((1, 6), (4, 4)), ((6, 1), (3, 7)), ((12, 12), (12, 12)),
                        ((37, 50), (64, 64)), ((24, 18), (12, 9)), ((50, 67), (33, 33)), ((33, 50), (65, 65))):
        img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        got = voc_dl.resize_bilinear_u8(img, out)
        ref = hsn_ref.cv2_resize_u8(img, (out[1], out[0]))
        assert got.dtype == np.uint8 and got.shape == ref.shape == (out[0], out[1], 3)
        assert np.array_equal(got, ref), ((H, W), out, np.abs(got.astype(int) - ref.astype(int)).max())
        if (H, W) == (2 * out[0], 2 * out[1]):
            v = img.astype(np.int64)
            assert np.array_equal(got, ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8))
            continue
        f = voc_dl.resize_bilinear_f64(img, out)
        assert np.abs(got.astype(np.float64) - f).max() <= 1.0 + 1e-9
        n_diff_round += int((got != np.clip(np.rint(f), 0, 255).astype(np.uint8)).sum())
        n_diff_trunc += int((got != f.astype(np.uint8)).sum())
    assert n_diff_round > 0 and n_diff_trunc > 0
    for v in (0, 1, 127, 254, 255):
        assert np.all(voc_dl.resize_bilinear_u8(np.full((9, 7, 3), v, np.uint8), (15, 4)) == v)
    img = rng.integers(0, 256, (5, 6, 3)).astype(np.uint8)
    same = voc_dl.resize_bilinear_u8(img, (5, 6))
    assert np.array_equal(same, img) and same is not img
    # the drivers' batch reader without a device: uint8 out, normalisation from the uint8 values (02_cues/utilities.py:177-180)
    from wsscam.cues import demo as cues_demo

    ims = [rng.integers(0, 256, (30, 40, 3)).astype(np.uint8), rng.integers(0, 256, (33, 33, 3)).astype(np.uint8)]
    norm, raw = cues_demo.read_batch(ims, (33, 33), [104, 117, 123], [255, 255, 255])
    assert raw.dtype == np.uint8 and np.array_equal(raw, hsn_ref.read_batch_u8(ims, (33, 33)))
    assert np.array_equal(norm, (raw - np.float64([104, 117, 123])) / 255.0)


def test_eval_cam_colour_label_readers(tmp_path):
    """Ground-truth readers of the ADP / DeepGlobe eval_cam branch (adp_semantic_segmentation_dataset.py:33-70,
    deepglobe_semantic_segmentation_dataset.py:21-64): split -> id list, colour PNG -> class index (an unlisted colour
    stays 0)."""
    import os

    from wsscam.step import eval_cam

    rng = np.random.default_rng(3)
    for htt, colours in eval_cam.ADP_CLS_COLOURS.items():
        lab = rng.integers(0, len(colours), (9, 11))
        rgb = np.asarray(colours, np.uint8)[lab]
        rgb[0, 0] = (1, 2, 3)  # not a class colour
        ref = lab.copy()
        ref[0, 0] = 0
        assert np.array_equal(eval_cam.label_from_colours(rgb, colours), ref)
    assert len(eval_cam.ADP_CLS_COLOURS["morph"]) == 29 and len(eval_cam.DEEPGLOBE_CLS_COLOURS) == 6
    os.makedirs(tmp_path / "ImageSets" / "Segmentation")
    os.makedirs(tmp_path / "SegmentationClassAug" / "ADP-func")
    (tmp_path / "ImageSets" / "Segmentation" / "segtest.txt").write_text("a\nb\n")
    (tmp_path / "ImageSets" / "Segmentation" / "train37.5.txt").write_text("c\n")
    lab = rng.integers(0, 5, (6, 7))
    _write(str(tmp_path / "SegmentationClassAug" / "ADP-func" / "b.png"), np.asarray(eval_cam.ADP_CLS_COLOURS["func"], np.uint8)[lab])
    ds = eval_cam.ADPSegLabels("evaluation", str(tmp_path), "func")
    assert ds.ids == ["a", "b"] and np.array_equal(ds.label(1), lab) and ds.label(1).dtype == np.uint8
    assert eval_cam.DeepGlobeSegLabels("train", str(tmp_path), is_balanced=True).ids == ["c"]
    with pytest.raises(ValueError):
        eval_cam.ADPSegLabels("val", str(tmp_path), "func")


def test_save_npy_object_loads_like_np_save(tmp_path):
    """The writer threads' .npy writer (wsscam.step.make_cam.save_npy_object) against np.save on the three dictionaries of
    make_cam.py:80-88: np.load(..., allow_pickle=True).item() -- what eval_cam.py:48, cam_to_ir_label.py:27 and
    make_sem_seg_labels.py:61 do -- gives the same keys, dtypes, shapes and values; the extension is appended like np.save does."""
    from wsscam.step.make_cam import save_npy_object

    rng = np.random.default_rng(3)
    cases = [
        {"keys": np.array([3, 7], np.int64), "cam": rng.random((2, 94, 125), dtype=np.float32),
         "high_res": rng.random((2, 375, 500), dtype=np.float32)},
        {"keys": np.array([0, 1, 5], np.int64), "cam": rng.random((3, 153, 153), dtype=np.float32)},   # DeepGlobe: no high_res
        {"keys": np.empty(0), "cam": np.empty(0), "high_res": np.empty(0)},                            # no class present
        {"keys": np.array([1], np.int64), "cam": rng.random((4, 20, 30), dtype=np.float32)[1:2, ::2]},  # a non-contiguous view
    ]
    for i, d in enumerate(cases):
        a, b = tmp_path / ("a%d" % i), tmp_path / ("b%d.npy" % i)
        save_npy_object(str(a), d)  # no extension given
        np.save(str(b), d)
        got = np.load(str(a) + ".npy", allow_pickle=True).item()
        ref = np.load(str(b), allow_pickle=True).item()
        assert list(got.keys()) == list(ref.keys())
        for k in ref:
            assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), (i, k)
    # round 6: dicts of one signature share the container's cached metadata pieces, written around the arrays by one C call
    # (wsc_host_write_segments).  A second dict of the same shapes -- other values, keys that are all zero bytes, a read-only
    # view like the ones cut out of a page-locked staging buffer -- must give the very bytes of the pickle.dump path.
    from wsscam.step import make_cam as mc

    for trial in range(3):
        hr = rng.random((2, 375, 500), dtype=np.float32)
        hr.flags.writeable = trial != 2
        d = {"keys": np.array([0, 0] if trial == 1 else [5, 19], np.int64), "cam": rng.random((2, 94, 125), dtype=np.float32),
             "high_res": hr}
        pth = tmp_path / ("c%d.npy" % trial)
        save_npy_object(str(pth), d)
        assert pth.read_bytes() == mc._npy_object_bytes(d), trial
        got = np.load(str(pth), allow_pickle=True).item()
        assert all(np.array_equal(got[k], d[k]) and got[k].dtype == d[k].dtype for k in d)
    lab = rng.integers(0, 21, (321, 321)).astype(np.uint8)
    mc.save_npy_array(str(tmp_path / "lab"), lab)
    np.save(str(tmp_path / "lab_ref.npy"), lab)
    assert (tmp_path / "lab.npy").read_bytes() == (tmp_path / "lab_ref.npy").read_bytes()
    sig = (("keys", "<i8", (2,), True), ("cam", "<f4", (2, 94, 125), True), ("high_res", "<f4", (2, 375, 500), True))
    assert mc._NPY_TEMPLATES.get(sig) is not None  # (the fast path was the one that ran)
