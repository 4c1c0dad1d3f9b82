"""GPU: the device HistoSegNet chain (csrc/hsn.hip: wsc_hsn_gradcam_post / _background / _cs_gradcam / _gather_unary)
against the numpy / scipy restatement of 03c_hsn/utilities.py:231-445 in oracle/hsn_ref.py.  fp32 on the device,
float64 in the oracle: stated tolerance 1e-5 absolute on maps that are O(1) (fp32 source coordinates and bilinear weights:
4e-6 measured at 40 -> 321), 2e-6 for the element-wise stages."""
import numpy as np
import pytest

from oracle import hsn_ref
from wsscam import _lib
from wsscam.hsn import demo as hsn_demo
from wsscam.hsn import utilities as hsn

pytestmark = pytest.mark.gpu


def _adp_like(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.full((H, W, 3), 235.0, np.float32) + rng.normal(0, 6, (H, W, 3))
    for _ in range(5):
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(0.1, 0.3) * min(H, W)
        a = 1 / (1 + np.exp(np.minimum((((yy - cy) ** 2 + (xx - cx) ** 2) / r ** 2 - 1) * 5, 50)))
        img = img * (1 - a[..., None]) + rng.uniform([150, 60, 120], [230, 140, 200]) * a[..., None]
    return np.clip(img, 0, 255).astype(np.uint8)


def test_hsn_gradcam_post_vs_oracle(ctx):
    rng = np.random.default_rng(1)
    for (B, h, C, S) in ((3, 10, 7, 33), (2, 40, 31, 321), (1, 1, 2, 5)):
        cams = rng.normal(0.1, 1.0, (B, h, h, C)).astype(np.float32)
        cams[0, :, :, 0] = -1.0  # a class that is negative everywhere: zero after max(., 0)
        scores = rng.uniform(0, 1, (B, C))
        is_pass = scores >= 0.4
        out_dev = ctx.alloc(B * C * S * S * 4)
        _lib.hsn_gradcam_post(ctx, ctx.to_device(cams), B, h, h, C, S, ctx.to_device((scores * is_pass).astype(np.float32)), out_dev)
        got = ctx.to_host(out_dev, (B, C, S, S), np.float32)
        ref = np.transpose(hsn_ref.grad_cam_post(cams, scores, is_pass, (S, S)), (0, 3, 1, 2))
        assert np.abs(got - ref).max() <= 1e-5, (B, h, C, S, np.abs(got - ref).max())
    # the numpy-signature mirror goes through the same kernels (needs a network: covered by test_hsn_segment_adp_driver)


def test_hsn_background_vs_scipy(ctx):
    import scipy.ndimage
    import scipy.special

    rng = np.random.default_rng(2)
    for (B, H, W) in ((2, 64, 48), (1, 321, 321), (1, 5, 3), (1, 1, 20)):
        rgb = np.stack([_adp_like(rng, H, W) for _ in range(B)])
        rgb[0, : H // 2, : W // 2] = 250  # a white (background) region
        bg_dev = ctx.alloc(B * H * W * 8)
        _lib.hsn_background(ctx, ctx.to_device(rgb), B, H, W, bg_dev)
        got = ctx.to_host(bg_dev, (B, H, W), np.float64)
        ref = np.stack([scipy.ndimage.gaussian_filter(0.75 * scipy.special.expit(4 * (np.mean(rgb[i], axis=-1) - 240)), sigma=2)
                        for i in range(B)])
        assert np.abs(got - ref).max() <= 1e-12 and np.all((got > 0) == (ref > 0)), ((B, H, W), np.abs(got - ref).max())


@pytest.mark.parametrize("htt", ["morph", "func"])
def test_hsn_modify_and_cs_gradcam_vs_oracle(ctx, htt):
    rng = np.random.default_rng(3 if htt == "morph" else 4)
    B, S = 2, 57
    N = S * S
    ac = hsn_demo.ADPClasses()
    C_all = len(ac.classes["all"])
    H = np.maximum(rng.normal(0.0, 0.4, (B, C_all, S, S)), 0).astype(np.float32)
    H[0, 5] = 0  # a class without mass
    H[:, :, :4, :4] = 0  # a corner where everything ties at zero (np.argmax's first-maximum rule)
    raw = np.stack([_adp_like(rng, S, S) for _ in range(B)])
    raw[1, 20:40, 10:50] = 252
    valid = ac.classes["valid_" + htt]
    Cv = len(valid)
    Y = np.zeros((B, Cv, S, S))
    Y[:, ac.classinds[htt + "2valid"]] = H[:, ac.classinds["all2" + htt]]
    adipose_all = [i for i, x in enumerate(ac.classes["all"]) if x in ["A.W", "A.B", "A.M"]]
    Yr = hsn_ref.modify_by_htt(Y, raw, valid, gradcam_adipose=H[:, adipose_all].astype(np.float64) if htt == "func" else None)
    csr = hsn_ref.get_cs_gradcam(Yr, valid, htt)
    # device, fused, from the gated Grad-CAM stack
    H_dev = ctx.to_device(H.reshape(B, C_all, N))
    bg_dev = ctx.alloc(B * N * 8)
    _lib.hsn_background(ctx, ctx.to_device(raw), B, S, S, bg_dev)
    src_of = [-1] * Cv
    for v, a in zip(ac.classinds[htt + "2valid"], ac.classinds["all2" + htt]):
        src_of[v] = a
    bg_ind, other_ind, ex = hsn._htt_tables(valid, htt == "func")
    cs_dev, y_dev, mass_dev = ctx.alloc(B * Cv * N * 4), ctx.alloc(B * Cv * N * 4), ctx.alloc(B * Cv * 4)
    _lib.hsn_cs_gradcam(ctx, H_dev, B, C_all, N, bg_dev, src_of, bg_ind, other_ind, ex, adipose_all if htt == "func" else None,
                        cs_dev, y_dev, mass_dev)
    y = ctx.to_host(y_dev, (B, Cv, S, S), np.float32)
    cs = ctx.to_host(cs_dev, (B, Cv, S, S), np.float32)
    mass = ctx.to_host(mass_dev, (B, Cv), np.uint32)
    assert np.abs(y - Yr).max() <= 2e-6
    # the margin map: identical arg-max except where the two largest values are closer than the fp32 resolution
    srt = np.sort(Yr, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-5
    tie = (srt[:, -1] - srt[:, -2]) == 0
    assert np.abs(cs - csr)[np.broadcast_to((safe | tie)[:, None], cs.shape)].max() <= 4e-6
    assert (safe | tie).mean() > 0.99
    for b in range(B):
        keep = hsn_ref.pass_classes(csr[b])
        assert list(np.nonzero(mass[b])[0]) == list(keep), (b, np.nonzero(mass[b])[0], keep)
    # numpy-signature mirrors (03c_hsn/utilities.py:306, :367) run the same kernels
    Ym = hsn.modify_by_htt(Y.copy(), raw, valid, gradcam_adipose=H[:, adipose_all] if htt == "func" else None, ctx=ctx)
    assert np.abs(Ym - Yr).max() <= 2e-6
    cm = hsn.get_cs_gradcam(Yr, valid, htt, ctx=ctx)
    assert np.abs(cm - csr)[np.broadcast_to((safe | tie)[:, None], cs.shape)].max() <= 4e-6
    # unaries of the passing classes (unary_from_softmax, utilities.py:431)
    keep = np.nonzero(mass[1])[0]
    u_dev = ctx.alloc(len(keep) * N * 4)
    _lib.hsn_gather_unary(ctx, cs_dev, [(1 * Cv + int(c)) * N for c in keep], N, u_dev)
    U = ctx.to_host(u_dev, (len(keep), N), np.float32)
    Ur = -np.log(np.clip(cs[1, keep].reshape(len(keep), N), 1e-5, 1.0))
    assert np.abs(U - Ur).max() <= 2e-6


def test_hsn_evaluation_tail_1088(ctx):
    """03c_hsn/demo.py:386-408, 424-428 (the '1088 x 1088' of BASELINE config 5): label maps at 321 x 321 -> cv2 INTER_NEAREST
    to 1088 x 1088 -> per-class confusion rows / intersections / unions / ground-truth counts / IoU / mIoU, on the device
    (wsc_label_confusion_nn through hsn.demo.LabelEvaluator) against the reference's loop restated in numpy, over two batches
    (the counters accumulate), with ground-truth pixels whose colour matches no class."""
    from wsscam.hsn import demo as hsn_demo
    from wsscam.step.eval_cam import ADP_CLS_COLOURS

    rng = np.random.default_rng(5)
    for htt in ("morph", "func"):
        colours = np.array(ADP_CLS_COLOURS[htt])
        n = len(colours)
        ev = hsn_demo.LabelEvaluator(ctx, colours)
        conf = np.zeros((n, n))
        inter, union, gtc = np.zeros(n), np.zeros(n), np.zeros(n)
        for batch in range(2):
            B = 2 + batch
            labs = [np.kron(rng.integers(0, n, (11, 11)), np.ones((30, 30), np.int64))[:321, :321] for _ in range(B)]
            gts = []
            for _ in range(B):
                gi = np.kron(rng.integers(0, n + 1, (17, 17)), np.ones((64, 64), np.int64))  # index n: a colour of no class
                pal = np.concatenate([colours, [[1, 2, 3]]], 0)
                gts.append(pal[gi].astype(np.uint8))
            preds = ev.update(labs, gts, want_pred=True)
            for b in range(B):
                # demo.py:392-405 verbatim (cv2.resize nearest: source index min(floor(x * (1. / (dst / src))), src - 1))
                sy = np.minimum(np.floor(np.arange(1088) * (1.0 / (1088 / 321))).astype(np.int64), 320)
                pred_idx = labs[b][sy][:, sy]
                assert np.array_equal(preds[b], pred_idx)
                gt_r, gt_g, gt_b = gts[b][:, :, 0], gts[b][:, :, 1], gts[b][:, :, 2]
                for k, c in enumerate(colours):
                    gt_mask = (gt_r == c[0]) & (gt_g == c[1]) & (gt_b == c[2])
                    pred_mask = pred_idx == k
                    conf[k, :] += np.bincount(pred_idx[gt_mask], minlength=n)
                    inter[k] += np.sum(gt_mask & pred_mask)
                    union[k] += np.sum(gt_mask | pred_mask)
                    gtc[k] += np.sum(gt_mask)
        m = ev.metrics()
        assert np.array_equal(m["confusion_matrix"], conf) and np.array_equal(m["intersects"], inter)
        assert np.array_equal(m["unions"], union) and np.array_equal(m["gt_count"], gtc)
        assert m["mIoU"] == float(np.mean(inter / (union + 1e-7)))


@pytest.mark.parametrize("dataset", ["VOC2012", "DeepGlobe"])
def test_eval_cues_vs_reference_loop(ctx, tmp_path, dataset):
    """02_cues/demo.py:323-484 (eval_cues): cues -> 41 x 41 arg-max map -> cv2 nearest resize to each ground truth's own size
    -> per-class intersections / unions over the set -> IoU, mIoU, metrics file.  The device counters (wsc_label_confusion_nn)
    against the reference's loop restated line by line in numpy: exact (integer counts), incl. VOC's void label 255, images
    without any cue and DeepGlobe's 'no class claims the pixel' label."""
    from tests.test_gpu_edge import _cv2_nearest
    from wsscam.cues import demo as cues_demo
    from wsscam.step.eval_cam import DEEPGLOBE_CLS_COLOURS

    rng = np.random.default_rng(61)
    voc = dataset == "VOC2012"
    n_cls = 21 if voc else 6
    sizes = [(50, 70), (41, 41), (120, 95), (33, 200), (64, 64)]
    cues, gts = {}, []
    for i, (H, W) in enumerate(sizes):
        one = np.zeros((n_cls, 41, 41), np.int64)
        if i != 3:  # image 3 has no cue at all
            lab = rng.integers(0, n_cls, (41, 41))
            keep = rng.random((41, 41)) < 0.6
            one[lab[keep], np.nonzero(keep)[0], np.nonzero(keep)[1]] = 1
        cues["%d_cues" % i] = np.array(np.where(one))
        cues["%d_labels" % i] = np.unique(cues["%d_cues" % i][0])
        if voc:
            g = rng.integers(0, n_cls, (H, W)).astype(np.uint8)
            g[rng.random((H, W)) < 0.1] = 255
            gts.append(g)
        else:
            cols = np.asarray(list(DEEPGLOBE_CLS_COLOURS) + [(0, 0, 0)], np.uint8)  # + the 'unknown' colour, which is not scored
            gts.append(cols[rng.integers(0, n_cls + 1, (H, W))])
    out = cues_demo.eval_cues(dataset, "VGG16", 0.2, 2, cues=cues, gts=gts, out_dir=str(tmp_path), ctx=ctx, is_verbose=False)
    inter, union = np.zeros(n_cls), np.zeros(n_cls)
    for i, g in enumerate(gts):
        ci = cues["%d_cues" % i]
        pred = np.zeros((41, 41, n_cls))
        pred[ci[1], ci[2], ci[0]] = 1.0
        pm = np.argmax(pred, axis=-1)
        if not voc:
            pm[np.sum(pred, axis=-1) == 0] = 6
        pidx = _cv2_nearest(np.uint8(pm), g.shape[:2])
        for k in range(n_cls):
            gm = (g == k) if voc else np.all(g == np.asarray(DEEPGLOBE_CLS_COLOURS[k], np.uint8)[None, None, :], axis=2)
            inter[k] += np.sum(gm & (pidx == k))
            union[k] += np.sum(gm | (pidx == k))
    assert np.array_equal(out["intersects"], inter) and np.array_equal(out["unions"], union)
    assert np.allclose(out["IoU"], inter / (union + 1e-7), rtol=0, atol=0) and out["mIoU"] == float(np.mean(inter / (union + 1e-7)))
    rows = open(tmp_path / ("metrics_%s_VGG16_%s.csv" % (dataset, "val" if voc else "test"))).read().strip().splitlines()
    assert rows[0] == ",Class,IoU" and len(rows) == n_cls + 2 and rows[-1].split(",")[1] == "Mean"
