"""CPU: host-side seed logic of 02_cues (utilities.py:183-278) and the localization_cues layout SEC/DSRG read."""
import pickle

import numpy as np
import pytest

from wsscam.cues import utilities as cues


def _brute(loc):
    """Independent per-pixel statement: the label of a pixel is the covering class with the smallest mask
    (visiting order = np.argsort(-area), later classes overwrite)."""
    B, C, H, W = loc.shape
    out = np.zeros((B, H, W), np.int64)
    for b in range(B):
        area = loc[b].sum(axis=(1, 2))
        order = np.argsort(-area)
        for y in range(H):
            for x in range(W):
                for c in order:
                    if loc[b, c, y, x]:
                        out[b, y, x] = c + 1
    return out


def test_fg_cues_layout_and_overlap():
    rng = np.random.default_rng(0)
    H_fg = rng.random((3, 4, 7, 6))
    H_fg[:, 1, :3] += 1.0  # a dominant region
    d = cues.get_fg_cues({}, H_fg, [np.array([1, 2]), np.array([0]), np.array([3])], [10, 11, 12], 0.5)
    assert sorted(d) == ["10_cues", "10_labels", "11_cues", "11_labels", "12_cues", "12_labels"]
    loc = np.stack([H_fg[:, c] > 0.5 * H_fg[:, c].max() for c in range(4)], 1).astype(np.int64)
    ref = _brute(loc)
    for i, x in enumerate([10, 11, 12]):
        c = d["%d_cues" % x]
        assert c.dtype == np.int64 and c.shape[0] == 3
        lab = np.zeros((7, 6), np.int64)
        lab[c[1], c[2]] = c[0] + 1
        assert np.array_equal(lab, ref[i])
    # round-trips through pickle like localization_cues.pickle (02_cues/demo.py:217-222)
    d2 = pickle.loads(pickle.dumps(d))
    assert np.array_equal(d2["11_cues"], d["11_cues"])


def test_fgbg_cues_background_channel():
    rng = np.random.default_rng(1)
    H_fg = rng.random((2, 3, 9, 9))
    H_bg = rng.random((2, 2, 9, 9))
    d = cues.get_fgbg_cues({}, H_fg, H_bg, [np.array([2]), np.array([1, 3])], [0, 1], 0.6)
    import scipy.ndimage

    for b in range(2):
        grad = scipy.ndimage.median_filter(H_bg[b].sum(0), 3)
        thr = np.sort(grad.ravel())[int(0.1 * 81)]
        c = d["%d_cues" % b]
        bg_pixels = set(zip(c[1][c[0] == 0], c[2][c[0] == 0]))
        # every background cue pixel is below the 10th-percentile threshold
        assert all(grad[y, x] < thr for (y, x) in bg_pixels)
        assert c[0].max() <= 3 and c[0].min() >= 0
    # batch-level max (Q7): scaling one image's activations changes the other image's cues
    H2 = H_fg.copy()
    H2[0] *= 10
    d2 = cues.get_fgbg_cues({}, H2, H_bg, [np.array([2]), np.array([1, 3])], [0, 1], 0.6)
    assert d2["1_cues"].shape[1] < d["1_cues"].shape[1]


def test_adp_update_cues_per_image_threshold():
    """02_cues/adp_cues.py:304-339: per-image, per-class max (Q7) -- scaling one image leaves the others' cues
    unchanged, unlike the VOC/DeepGlobe variant."""
    rng = np.random.default_rng(5)
    g = rng.random((3, 5, 8, 8))
    inds = [np.array([0, 2]), np.array([1]), np.array([3, 4])]
    d = cues.update_cues_adp({}, g, inds, [7, 8, 9], 0.7)
    loc = (g > 0.7 * g.max(axis=(2, 3), keepdims=True)).astype(np.int64)
    ref = _brute(loc)
    for i, x in enumerate([7, 8, 9]):
        c = d["%d_cues" % x]
        lab = np.zeros((8, 8), np.int64)
        lab[c[1], c[2]] = c[0] + 1
        assert np.array_equal(lab, ref[i])
        assert np.array_equal(d["%d_labels" % x], inds[i])
    g2 = g.copy()
    g2[0] *= 10
    d2 = cues.update_cues_adp({}, g2, inds, [7, 8, 9], 0.7)
    assert all(np.array_equal(d2["%d_cues" % x], d["%d_cues" % x]) for x in (7, 8, 9))


def test_eval_sem_seg_report(tmp_path):
    """step.eval_sem_seg.run: confusion / IoU / CSV / log lines from label PNGs (eval_sem_seg.py:12-64)."""
    import types

    from PIL import Image

    from wsscam.step import eval_sem_seg

    rng = np.random.default_rng(8)
    names = ["2007_000033", "2007_000042"]
    seg_dir = tmp_path / "seg"
    seg_dir.mkdir()
    gts, preds = {}, {}
    for n in names:
        gt = rng.integers(0, 4, (20, 30))
        gt[rng.random((20, 30)) < 0.1] = 255
        pr = np.where(rng.random((20, 30)) < 0.7, np.where(gt == 255, 0, gt), rng.integers(0, 4, (20, 30))).astype(np.uint8)
        gts[n], preds[n] = gt, pr
        Image.fromarray(pr).save(seg_dir / (n + ".png"))
    args = types.SimpleNamespace(dataset="voc12", ids=names, gt_labels=gts, sem_seg_out_dir=str(seg_dir),
                                 class_names={"bg": ["background"], "fg": ["a", "b", "c"]}, eval_dir=str(tmp_path / "eval"),
                                 run_name="run", split="val", logfile=str(tmp_path / "log.txt"))
    out = eval_sem_seg.run(args)
    conf = np.zeros((4, 4), np.int64)
    for n in names:
        m = gts[n] != 255
        np.add.at(conf, (gts[n][m], preds[n][m]), 1)
    assert np.array_equal(out["confusion"], conf)
    iou = np.diag(conf) / (conf.sum(0) + conf.sum(1) - np.diag(conf))
    assert np.allclose(out["iou"], iou) and abs(out["miou"] - iou.mean()) < 1e-12
    rows = open(tmp_path / "eval" / "run_val_iou.csv").read().strip().split("\n")
    assert rows[0] == ",iou" and [r.split(",")[0] for r in rows[1:]] == ["background", "a", "b", "c", "miou"]
    log = open(tmp_path / "log.txt").read()
    assert "[eval_sem_seg, val] miou: " + str(np.nanmean(iou)) in log


def test_grad_cam_alpha_closed_forms_vs_autograd():
    """get_grad_cam_weights (02_cues/utilities.py:60-99, common_cnn.py:84-121) for BOTH classifier heads against
    torch.autograd on the restated nets (oracle/cnn_ref.grad_cam_weights, which differentiates with respect to the final
    Activation's output = the PRE-BatchNorm tensor, so the gradient carries gamma / sqrt(var + eps) per channel):
    GAP + Linear (VGG16) and MaxPool + global max + Linear (M7 -- the all-zeros image ties every position; alpha does not
    depend on the winner), incl. an odd feature-map size where MaxPool2d(2, 2) drops the last row / column, negative
    BatchNorm scales (the pooled maximum then sits on the activation's minimum) and a stack without BatchNorm."""
    from oracle import cnn_ref
    from wsscam.net import common

    C = 7
    cases = (("vgg16", cnn_ref.VGG16_CFG, 64, True), ("m7", cnn_ref.M7_CFG, 32, True), ("m7", cnn_ref.M7_CFG, 28, True),
             ("vgg16", cnn_ref.VGG16_CFG, 32, False))
    for root, cfg, S, batchnorm in cases:
        sd = cnn_ref.make_plain_state_dict(root, cfg, C, batchnorm, seed=11)
        bn = cnn_ref.last_bn_key(sd, root, cfg)
        assert (bn is not None) == batchnorm
        if bn:
            sd[bn + ".weight"][::3] *= -1.0

        class Model:  # what cues.get_grad_cam_weights reads from a wsscam CAM wrapper
            _sd = {k: v.numpy() for k, v in sd.items()}

        Model.root = root
        ref = cnn_ref.grad_cam_weights(sd, root, cfg, S, C)
        alpha = cues.get_grad_cam_weights(Model, cues.find_final_layer(Model), np.zeros((1, S, S, 3), np.float32))
        assert alpha.shape == ref.shape
        assert np.abs(alpha - ref).max() <= 2e-5 * np.abs(ref).max(), (root, S, np.abs(alpha - ref).max())
        h = S // (8 if root == "vgg16" else 4)
        W = Model._sd[root + ".classifier.0.weight"]
        raw = common.grad_cam_alpha(W, h, h, "max" if root == "m7" else "avg", False)
        assert np.allclose(raw, W.T / (h * h))
        affine = common.last_bn_affine(Model._sd, root)
        if batchnorm:
            g, v = Model._sd[bn + ".weight"].astype(np.float64), Model._sd[bn + ".running_var"].astype(np.float64)
            assert np.allclose(affine[0], g / np.sqrt(v + 1e-3)) and (affine[0] < 0).any()
            # without the BatchNorm fold the closed form is NOT the reference's alpha (the round-2 behaviour)
            plain = common.grad_cam_alpha(W, h, h, "max" if root == "m7" else "avg", True)
            assert np.abs(plain - ref).max() > 0.05 * np.abs(ref).max()
        else:
            assert affine is None


def test_pre_bn_head_algebra():
    """net.common.pre_bn_head: a head on the post-BatchNorm map that equals einsum(pre-BN activation, alpha)."""
    from wsscam.net import common

    rng = np.random.default_rng(5)
    F, C = 12, 4
    pre = rng.random((3, 5, 5, F))
    s, t = rng.uniform(-2, 2, F), rng.normal(size=F)
    alpha = rng.normal(size=(F, C))
    w, b = common.pre_bn_head(alpha, (s, t))
    post = pre * s + t
    assert np.allclose(np.einsum("ijkl,lm->ijkm", post, w) + b, np.einsum("ijkl,lm->ijkm", pre, alpha))
    w0, b0 = common.pre_bn_head(alpha, None)
    assert np.array_equal(w0, alpha) and not b0.any()
    s[3] = 0
    with pytest.raises(ValueError):
        common.pre_bn_head(alpha, (s, t))


def test_keras_store_settings_sessions_and_csv(tmp_path, monkeypatch):
    """keras_store: settings.ini keys (02_cues/demo.py:16-21), the background session's directory rule (:139-150) and the
    split CSV reader (02_cues/dataset.py:98-124)."""
    import os

    from wsscam import keras_store as ks

    ini = tmp_path / "settings.ini"
    ini.write_text("[Download Directory]\ndata_dir = ../database\n\n[Data Folders]\nmodel_cnn_dir = models_cnn\ncues_dir = cues\n")
    work = tmp_path / "02_cues"
    work.mkdir()
    monkeypatch.chdir(work)
    st = ks.read_settings()  # '../settings.ini' relative to the working directory, like the reference
    assert st["DATA_ROOT"] == str(tmp_path / "database") and st["MODEL_ROOT"] == str(tmp_path / "database" / "models_cnn")
    assert st["CUES_ROOT"] == str(tmp_path / "database" / "cues")
    s = ks.fgbg_sessions("/m/VOC2012_VGG16", "VOC2012_VGG16", ["fg", "bg"])
    assert s == {"fg": ("/m/VOC2012_VGG16", "VOC2012_VGG16"), "bg": ("/m/VOC2012_VGG16bg", "VOC2012_VGG16bg")}
    assert ks.fgbg_sessions("/m/X_fg", "X_fg", ["bg"]) == {"bg": ("/m/X_bg", "X_bg")}
    voc = tmp_path / "database" / "VOCdevkit" / "VOC2012"
    os.makedirs(voc / "ImageSets" / "Segmentation")
    head = "Patch Names," + ",".join(ks.VOC_CLASSES)
    for split, n in (("trainaug", 3), ("val", 2)):
        rows = ["%s_%d.jpg,%s" % (split, i, ",".join("1" if c == i else "0" for c in range(20))) for i in range(n)]
        (voc / "ImageSets" / "Segmentation" / (split + ".csv")).write_text(head + "\n" + "\n".join(rows) + "\n")
    ds = ks.Dataset("VOC2012", 321, 4)  # database_dir defaults to <parent of cwd>/database
    assert ds.sets == ["trainaug", "val"] and ds.is_evals == [False, True] and len(ds.class_names) == 20
    g = ds.set_gens["val"]
    assert g.filenames == ["val_0.jpg", "val_1.jpg"] and g.directory == str(voc / "JPEGImages")
    assert g.data.shape == (2, 20) and g.data[1, 1] == 1 and g.data.sum() == 2
    with pytest.raises(FileNotFoundError):
        ks.read_settings(str(tmp_path / "missing.ini"))
    # 03c_hsn/dataset.py:57-59: VOC2012 lives under VOC_trainaug_val, evaluation split only, database_dir = DATA_ROOT argument
    voc_h = tmp_path / "database" / "VOCdevkit" / "VOC_trainaug_val" / "VOC2012"
    os.makedirs(voc_h / "ImageSets" / "Segmentation")
    (voc_h / "ImageSets" / "Segmentation" / "val.csv").write_text(head + "\nv0.jpg," + ",".join(["0"] * 19 + ["1"]) + "\n")
    dh = ks.Dataset("VOC2012", 321, 4, database_dir=st["DATA_ROOT"], layout="hsn")
    assert dh.sets == ["val"] and dh.is_evals == [True] and dh.set_gens["val"].filenames == ["v0.jpg"]
    assert dh.set_gens["val"].directory == str(voc_h / "JPEGImages") and dh.set_gens["val"].data[0, 19] == 1
    with pytest.raises(ValueError):
        ks.Dataset("VOC2012", 321, 4, layout="hsn")  # no default directory in the HSN class
    with pytest.raises(ValueError):
        ks.Dataset("DeepGlobe_train75", 224, 4, database_dir=st["DATA_ROOT"], layout="hsn")  # a 02_cues name


def test_adp_adipose_channels_follow_the_reference():
    """03c_hsn/demo.py:347-371 (twin: 02_cues/demo.py:296-310): the functional 'Other' channel takes
    `Y_gradcam['morph'][:, adipose_inds]` with adipose_inds = positions of A.W / A.B / A.M in classes['morph'] -- but
    Y_gradcam['morph'] is the VALID stack (channel 0 = Background), so the maps are S.R, A.W, A.B.  The device driver reads
    the all-class stack directly: its channel list must select exactly the maps the reference's expression selects."""
    from oracle import hsn_ref
    from wsscam.hsn import demo as hsn_demo

    rng = np.random.default_rng(4)
    for all_classes in (None, list(reversed(hsn_ref.ADP_MORPH + hsn_ref.ADP_FUNC))):
        classes, inds = hsn_ref.adp_class_tables(all_classes)
        H = rng.random((2, len(classes["all"]), 3, 3))
        Y = np.zeros((2, len(classes["valid_morph"]), 3, 3))            # demo.py:353-355
        Y[:, inds["morph2valid"]] = H[:, inds["all2morph"]]
        adipose_inds = [i for i, x in enumerate(classes["morph"]) if x in ["A.W", "A.B", "A.M"]]  # demo.py:368
        ref = Y[:, adipose_inds]                                        # demo.py:369
        got = H[:, hsn_demo.adipose_source_channels(hsn_demo.ADPClasses(all_classes))]
        assert np.array_equal(got, ref)
        if all_classes is None:
            assert [classes["all"][c] for c in hsn_demo.adipose_source_channels(hsn_demo.ADPClasses())] == ["S.R", "A.W", "A.B"]
    ac = hsn_demo.ADPClasses()
    assert (ac.classes, ac.classinds) == hsn_ref.adp_class_tables()
