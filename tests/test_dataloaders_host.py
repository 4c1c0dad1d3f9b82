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
