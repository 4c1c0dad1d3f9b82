"""GPU: permutohedral dense-CRF (wsc_crf_*) against the C restatement oracle/densecrf_ref.c.

Stated tolerances: arg-max label agreement >= 99.5 %, max|dQ| <= 1e-3.  The
lattice build is integer/index work and must match the oracle exactly: identical vertex counts.
"""
import numpy as np
import pytest

from tests import helpers
from wsscam import _lib
from wsscam.hsn import utilities as hsn_utilities
from wsscam.misc import imutils

pytestmark = pytest.mark.gpu

CFGS = [(1.5, 3, 40, 13, 10, 10), (3, 3, 50, 5, 10, 10), (3 / 12, 3, 80 / 12, 13, 10, 5)]


def _gpu_crf(ctx, rgbs, Us, cfg):
    B = len(rgbs)
    H, W, _ = rgbs[0].shape
    M = Us[0].shape[0]
    rgb_dev = ctx.to_device(np.stack(rgbs))
    u_dev = ctx.to_device(np.stack(Us))
    q_dev = ctx.alloc(B * M * H * W * 4)
    a_dev = ctx.alloc(B * H * W * 4)
    crf = _lib.Crf(ctx, rgb_dev, B, H, W, cfg[0], cfg[2], cfg[3])
    vg, vb = crf.lattice_sizes()
    _gpu_crf.on_chip = crf.gaussian_on_chip(M)
    crf.inference(u_dev, M, cfg[1], cfg[4], int(cfg[5]), q_dev, a_dev)
    q = ctx.to_host(q_dev, (B, M, H * W), np.float32)
    a = ctx.to_host(a_dev, (B, H * W), np.int32)
    crf.close()
    return q, a, vg, vb


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("M", [1, 2, 3, 6, 21, 32])
def test_crf_vs_oracle(ctx, cfg, M):
    rng = np.random.default_rng(100 + M)
    H, W = 57, 75
    cases = [helpers.synth_crf_case(rng, H, W, M) for _ in range(3)]
    q, a, vg, vb = _gpu_crf(ctx, [c[0] for c in cases], [c[1] for c in cases], cfg)
    for b, (rgb, U, _) in enumerate(cases):
        qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
        assert (vg[b], vb[b]) == (ls[0], ls[1]), "lattice vertex counts differ: %s vs %s" % ((vg[b], vb[b]), ls)
        assert np.abs(q[b] - qr).max() <= 1e-3, np.abs(q[b] - qr).max()
        assert (a[b] == ar).mean() >= 0.995
        assert np.abs(q[b].sum(0) - 1).max() <= 1e-5
        assert np.array_equal(a[b], q[b].argmax(0))


def test_crf_321_config3(ctx):
    """BASELINE config 3 size: 321x321, M=21, 10 iterations, one image (oracle takes ~1 s)."""
    rng = np.random.default_rng(7)
    rgb, U, _ = helpers.synth_crf_case(rng, 321, 321, 21)
    cfg = (1.5, 3, 40, 13, 10, 10)
    q, a, vg, vb = _gpu_crf(ctx, [rgb], [U], cfg)
    qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
    assert (vg[0], vb[0]) == (ls[0], ls[1])
    assert np.abs(q[0] - qr).max() <= 1e-3
    assert (a[0] == ar).mean() >= 0.995


def test_crf_invariants(ctx):
    rng = np.random.default_rng(8)
    rgb, U, _ = helpers.synth_crf_case(rng, 40, 52, 5)
    sm = np.exp(-U - (-U).max(0, keepdims=True))
    sm /= sm.sum(0, keepdims=True)
    q, _, _, _ = _gpu_crf(ctx, [rgb], [U], (1.5, 0, 40, 13, 0, 5))     # zero compat
    assert np.abs(q[0] - sm).max() <= 1e-6
    q, _, _, _ = _gpu_crf(ctx, [rgb], [U], (1.5, 3, 40, 13, 10, 0))    # zero iterations
    assert np.abs(q[0] - sm).max() <= 1e-6
    # a batch is the same as its images one by one (no cross-image leakage through the shared rows)
    rgb2, U2, _ = helpers.synth_crf_case(rng, 40, 52, 5)
    cfg = (3, 3, 50, 5, 10, 4)
    qb, ab, _, _ = _gpu_crf(ctx, [rgb, rgb2], [U, U2], cfg)
    q1, a1, _, _ = _gpu_crf(ctx, [rgb], [U], cfg)
    q2, a2, _, _ = _gpu_crf(ctx, [rgb2], [U2], cfg)
    assert np.array_equal(qb[0], q1[0]) and np.array_equal(qb[1], q2[0])
    # run-to-run bit reproducibility (sorted splat lists, no float atomics)
    qb2, _, _, _ = _gpu_crf(ctx, [rgb, rgb2], [U, U2], cfg)
    assert np.array_equal(qb, qb2)


def test_crf_shared_gaussian_lattice_cache():
    """The Gaussian lattice is built once per (H, W, sxy) and ctx and shared by every image of a batch:
    a ctx that has served other sizes / widths / batch sizes gives bit-identical results to a fresh one."""
    rng = np.random.default_rng(21)
    M = 4
    small = [helpers.synth_crf_case(rng, 33, 47, M) for _ in range(3)]
    other = [helpers.synth_crf_case(rng, 47, 33, M) for _ in range(2)]
    cfg_a, cfg_b = (1.5, 3, 40, 13, 10, 3), (3, 3, 50, 5, 10, 3)

    def run(c, cases, cfg):
        return _gpu_crf(c, [k[0] for k in cases], [k[1] for k in cases], cfg)

    fresh = {}
    for name, cases, cfg in (("a3", small, cfg_a), ("b3", small, cfg_b), ("a1", small[1:2], cfg_a),
                             ("o2", other, cfg_a)):
        c = _lib.Context(0)
        fresh[name] = run(c, cases, cfg)
        c.close()
    c = _lib.Context(0)
    for name, cases, cfg in (("a3", small, cfg_a), ("o2", other, cfg_a), ("b3", small, cfg_b),
                             ("a1", small[1:2], cfg_a), ("a3", small, cfg_a), ("o2", other, cfg_a)):
        q, a, vg, vb = run(c, cases, cfg)
        assert np.array_equal(q, fresh[name][0]) and np.array_equal(a, fresh[name][1]), name
        assert list(vg) == list(fresh[name][2]) and list(vb) == list(fresh[name][3])
        assert len(set(vg)) == 1  # position-only lattice: same vertex count for every image
    c.close()
    qr, ar, ls = helpers.crf_oracle(small[2][0], small[2][1], cfg_a)
    assert fresh["a3"][2][2] == ls[0] and np.abs(fresh["a3"][0][2] - qr).max() <= 1e-3


def test_crf_flat_image_long_rows(ctx):
    """A constant image puts thousands of pixels on one bilateral vertex (long splat rows)."""
    H, W, M = 96, 96, 3
    rgb = np.full((H, W, 3), 200, np.uint8)
    rng = np.random.default_rng(12)
    _, U, _ = helpers.synth_crf_case(rng, H, W, M)
    cfg = (3, 3, 80, 13, 10, 5)
    q, a, vg, vb = _gpu_crf(ctx, [rgb], [U], cfg)
    qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
    assert (vg[0], vb[0]) == (ls[0], ls[1])
    assert np.abs(q[0] - qr).max() <= 1e-3 and (a[0] == ar).mean() >= 0.995


def test_dcrf_process_mirror(ctx):
    """03c_hsn/utilities.py:399-445 semantics: per-image pass classes, scatter back, argmax over all."""
    rng = np.random.default_rng(13)
    B, C, H, W = 3, 6, 33, 41
    probs = np.zeros((B, C, H, W))
    imgs = np.zeros((B, H, W, 3), np.uint8)
    pass_sets = [[0, 2, 5], [1, 3, 4], [0, 2, 5]]
    for b in range(B):
        rgb, _, p = helpers.synth_crf_case(rng, H, W, 3)
        imgs[b] = rgb
        probs[b, pass_sets[b]] = p
    cfg = [1.5, 3, 40, 13, 10, 10]
    out = hsn_utilities.dcrf_process(probs, imgs, cfg, ctx=ctx)
    assert out.shape == (B, H, W) and out.dtype == np.int64
    for b in range(B):
        U = imutils.unary_from_softmax(probs[b, pass_sets[b]])
        qr, ar, _ = helpers.crf_oracle(imgs[b], U, cfg)
        ref = np.asarray(pass_sets[b])[ar].reshape(H, W)
        assert (out[b] == ref).mean() >= 0.995


def test_crf_inference_label_mirror(ctx):
    rng = np.random.default_rng(14)
    H, W, n_labels = 45, 38, 4
    rgb, _, p = helpers.synth_crf_case(rng, H, W, n_labels)
    labels = p.argmax(0)
    out = imutils.crf_inference_label(rgb, labels, "voc12", n_labels=n_labels, ctx=ctx)
    U = imutils.unary_from_labels(labels, n_labels, 0.7, zero_unsure=False)
    _, ar, _ = helpers.crf_oracle(rgb, U, (3, 3, 50, 5, 10, 10))
    assert out.shape == (H, W) and (out.reshape(-1) == ar).mean() >= 0.995


def test_crf_key_range_error(ctx):
    """Bilateral kernel widths so small that lattice coordinates leave the 12-bit packed range."""
    rgb = np.zeros((64, 64, 3), np.uint8)
    rgb[..., 0] = 255
    with pytest.raises(_lib.WscError) as ei:
        _lib.Crf(ctx, ctx.to_device(rgb), 1, 64, 64, 1.0, 0.02, 0.05)
    assert ei.value.status == _lib.WSC_ERR_KEY_RANGE


def test_cam_to_ir_label_mirror(ctx):
    """03b_irn/step/cam_to_ir_label.py:42-58 (VOC branch): two label-unary CRF runs on one lattice pair."""
    from wsscam.step import cam_to_ir_label

    rng = np.random.default_rng(16)
    H, W = 47, 59
    rgb, _, p = helpers.synth_crf_case(rng, H, W, 3)
    hi = (p[:2] / p[:2].max(axis=(1, 2), keepdims=True)).astype(np.float32)  # two max-normalised "CAMs"
    cam_dict = {"keys": np.array([4, 11]), "cam": None, "high_res": hi}
    out = cam_to_ir_label.ir_label_voc12(rgb, cam_dict, 0.30, 0.05, ctx=ctx)
    keys = np.array([0, 5, 12])

    def oracle_conf(thres):
        lab = np.argmax(np.pad(hi, ((1, 0), (0, 0), (0, 0)), mode="constant", constant_values=thres), axis=0)
        U = imutils.unary_from_labels(lab, 3, 0.7, zero_unsure=False)
        _, ar, _ = helpers.crf_oracle(rgb, U, (3, 3, 50, 5, 10, 10))
        return keys[ar.reshape(H, W)]

    fg, bg = oracle_conf(0.30), oracle_conf(0.05)
    ref = fg.copy()
    ref[fg == 0] = 255
    ref[bg + fg == 0] = 0
    assert out.shape == (H, W) and out.dtype == np.uint8
    assert (out == ref).mean() >= 0.995
    assert set(np.unique(out)) <= {0, 5, 12, 255}


def test_crf_random_sweep(ctx):
    """Seeded random sweep: image sizes 5..90 (non-square), M 1..32, batches of 1..4, all three configurations and
    iteration counts 1..10 -- identical lattices, max|dQ| <= 1e-3, label agreement >= 99.5 % on every image."""
    rng = np.random.default_rng(77)
    for it in range(10):
        H, W = int(rng.integers(5, 90)), int(rng.integers(5, 90))
        M = int(rng.integers(1, 33))
        B = int(rng.integers(1, 5))
        base = CFGS[it % len(CFGS)]
        cfg = base[:5] + (int(rng.integers(1, 11)),)
        cases = [helpers.synth_crf_case(rng, H, W, M) for _ in range(B)]
        q, a, vg, vb = _gpu_crf(ctx, [c[0] for c in cases], [c[1] for c in cases], cfg)
        for b, (rgb, U, _) in enumerate(cases):
            qr, ar, ls = helpers.crf_oracle(rgb, U, cfg)
            assert (vg[b], vb[b]) == (ls[0], ls[1]), (it, H, W, M, B)
            assert np.abs(q[b] - qr).max() <= 1e-3, (it, H, W, M, B, cfg, np.abs(q[b] - qr).max())
            assert (a[b] == ar).mean() >= 0.995, (it, H, W, M, B, (a[b] == ar).mean())


@pytest.mark.parametrize("case", [(47, 61, 5, 2, (1.5, 3, 40, 13, 10, 5)), (96, 130, 21, 3, (3, 3, 50, 5, 10, 4)),
                                  (9, 7, 2, 1, (3 / 12, 3, 80 / 12, 13, 10, 3)), (33, 200, 29, 1, (5, 3, 40, 13, 10, 2))])
def test_crf_fused_gaussian_blur_is_bit_identical(ctx, case):
    """The tiled three-pass blur of the Gaussian lattice (blur3_tile_kernel: one read + one write of the rows) against
    the three separate blur4 passes (ctx option OPT_CRF_FUSED_BLUR = 0): identical bits, on sizes whose lattices
    have ragged edge tiles, several replicas, wide (sxy = 5) and narrow (sxy = 0.25) kernels, M = 29 (LP = 8: two
    groups of float4s per row in the fused kernel)."""
    H, W, M, B, cfg = case
    rng = np.random.default_rng(H * 7 + W)
    rgbs, Us = [], []
    for _ in range(B):
        rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
        rgbs.append(rgb)
        Us.append(U)
    with ctx.option(_lib.OPT_CRF_GAUSS_ON_CHIP, 0):  # the blur kernels under test only run when the update kernel leaves the blur to them
        with ctx.option(_lib.OPT_CRF_FUSED_BLUR, 0):
            q_ref, a_ref, vg, _ = _gpu_crf(ctx, rgbs, Us, cfg)
        q, a, vg2, _ = _gpu_crf(ctx, rgbs, Us, cfg)
    assert list(vg) == list(vg2)
    assert np.array_equal(q, q_ref) and np.array_equal(a, a_ref)


@pytest.mark.parametrize("case", [(47, 61, 5, 2, (1.5, 3, 40, 13, 10, 5)), (96, 130, 21, 3, (3, 3, 50, 5, 10, 4)),
                                  (9, 7, 2, 1, (3 / 12, 3, 80 / 12, 13, 10, 3)), (33, 200, 29, 1, (5, 3, 40, 13, 10, 2)),
                                  (321, 321, 21, 2, (1.5, 3, 40, 13, 10, 3)), (41, 41, 21, 4, (3 / 12, 3, 80 / 12, 13, 10, 5)),
                                  (64, 80, 1, 2, (1.5, 3, 40, 13, 10, 2)), (100, 75, 12, 1, (0.7, 3, 30, 10, 10, 3))])
def test_crf_gaussian_blur_inside_update_is_bit_identical(ctx, case):
    """update_splat_kernel<.., GF>: the Gaussian lattice summed from its slot partials, blurred and sliced inside the update
    kernel (per-tile closed vertex sets, LDS) against the separate blur kernel + value-row gathers (ctx option OPT_CRF_GAUSS_ON_CHIP = 0): identical Q bits and labels -- one image and several, ragged edge tiles, wide / narrow kernels (the narrow ones
    have vertex sets too large for the LDS at M = 21 and silently take the unfused path), LP = 1 ... 8, and the labels-only
    call whose last update writes the arg-max."""
    H, W, M, B, cfg = case
    rng = np.random.default_rng(H * 11 + W)
    rgbs, Us = [], []
    for _ in range(B):
        rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
        rgbs.append(rgb)
        Us.append(U)
    with ctx.option(_lib.OPT_CRF_GAUSS_ON_CHIP, 0):
        q_ref, a_ref, vg, vb = _gpu_crf(ctx, rgbs, Us, cfg)
        assert not _gpu_crf.on_chip
    q, a, vg2, vb2 = _gpu_crf(ctx, rgbs, Us, cfg)
    if cfg[0] >= 1.0:  # (narrow kernels: on chip only while a tile's vertex set fits the LDS -- the 9 x 7 image does)
        assert _gpu_crf.on_chip, cfg
    print("gaussian on chip:", case, _gpu_crf.on_chip)
    assert list(vg) == list(vg2) and list(vb) == list(vb2)
    assert np.array_equal(q, q_ref) and np.array_equal(a, a_ref)
    # round 6: the default path forms the message INSIDE the update kernel (E = -U + message never leaves the Q stage);
    # OPT_CRF_MSG_IN_UPDATE = 0 is round 3's two-launch form (gauss_msg_kernel writes E, the update reads it): identical bits
    with ctx.option(_lib.OPT_CRF_MSG_IN_UPDATE, 0):
        q2, a2, _, _ = _gpu_crf(ctx, rgbs, Us, cfg)
    assert np.array_equal(q2, q_ref) and np.array_equal(a2, a_ref)


@pytest.mark.parametrize("case", [(321, 321, 21, 8, (1.5, 3, 40, 13, 10, 4)), (47, 61, 5, 1, (1.5, 3, 40, 13, 10, 3)),
                                  (96, 130, 21, 3, (3, 3, 50, 5, 10, 5)), (18, 23, 2, 37, (1.5, 3, 20, 10, 10, 2)),
                                  (70, 64, 3, 2, (3, 3, 6, 1.5, 10, 2)), (5, 4, 3, 300, (1.5, 3, 20, 10, 10, 2)),
                                  (300, 400, 4, 2, (3, 3, 10, 3, 10, 2)), (321, 321, 29, 2, (1.5, 3, 20, 6, 10, 2))])
def test_crf_bilateral_blur_on_chip_is_bit_identical(ctx, case):
    """blur_lds_kernel: the six passes of the bilateral lattice in one launch, one image x GW classes per workgroup as float
    planes in LDS, against one blur4_kernel launch per pass (ctx option OPT_CRF_BLUR_ON_CHIP = 0): identical Q bits and labels.
    Cases: the VOC size (3 classes per workgroup, 13 rows per thread), small images (4 classes, 4 rows), images of unequal
    vertex counts, more workgroups than CUs (several rounds), noisy / narrow-kernel images whose ~20-60 k vertices leave
    room for 1-2 classes per workgroup (40 rows per thread), M = 29."""
    H, W, M, B, cfg = case
    rng = np.random.default_rng(H * 17 + W)
    rgbs, Us = [], []
    for _ in range(B):
        rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
        rgbs.append(rgb)
        Us.append(U)
    with ctx.option(_lib.OPT_CRF_BLUR_ON_CHIP, 0):
        q_ref, a_ref, vg, vb = _gpu_crf(ctx, rgbs, Us, cfg)
    q, a, vg2, vb2 = _gpu_crf(ctx, rgbs, Us, cfg)
    print("bilateral vertices per image:", list(vb)[:4])
    assert list(vg) == list(vg2) and list(vb) == list(vb2)
    assert np.array_equal(q, q_ref) and np.array_equal(a, a_ref)


@pytest.mark.parametrize("case", [(47, 61, 5, 2, (1.5, 3, 40, 13, 10, 3)), (96, 130, 21, 3, (3, 3, 50, 5, 10, 2)),
                                  (33, 200, 4, 1, (5, 3, 40, 13, 10, 2)), (70, 64, 3, 2, (3, 3, 6, 1.5, 10, 2))])
def test_crf_lattice_build_rank_paths_agree(ctx, case):
    """Lattice build, stable rank of a tile's entries inside their vertex group: the pixel-mask path (a group holds one entry
    per pixel: rank = popcount of the lower pixels) against the ballot-matching walk (ctx option OPT_CRF_RANK_BALLOT = 1).
    Same lattice sizes, same Q bits.  The last case (bilateral sxy = 6, srgb = 1.5 on a noisy image) has tiles with more
    distinct vertices than the masks hold, so the default build itself takes the ballot path there."""
    H, W, M, B, cfg = case
    rng = np.random.default_rng(H * 13 + W)
    rgbs, Us = [], []
    for _ in range(B):
        rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
        rgbs.append(rgb)
        Us.append(U)
    with ctx.option(_lib.OPT_CRF_RANK_BALLOT, 1):
        q_ref, a_ref, vg, vb = _gpu_crf(ctx, rgbs, Us, cfg)
    q, a, vg2, vb2 = _gpu_crf(ctx, rgbs, Us, cfg)
    assert list(vg) == list(vg2) and list(vb) == list(vb2)
    assert np.array_equal(q, q_ref) and np.array_equal(a, a_ref)
    if cfg[3] == 1.5:
        assert max(vb) > 2.5 * H * W  # nearly every pixel owns its vertices: > 640 groups per 16 x 16 tile
    # the tile pass on the 2048-slot LDS table for every tile (one launch) against the default: 512 slots first, the tiles
    # with more than 384 distinct vertices (all of them in the noisy case) redone by the full-table launch
    with ctx.option(_lib.OPT_CRF_EMBED_FULL, 1):
        q3, a3, vg3, vb3 = _gpu_crf(ctx, rgbs, Us, cfg)
    assert list(vg) == list(vg3) and list(vb) == list(vb3)
    assert np.array_equal(q, q3) and np.array_equal(a, a3)


def test_crf_labels_only_call_matches_full_call(ctx):
    """q_dev = NULL (labels only) returns the labels of the full call (Q + arg max): first maximum in class order, also
    on exact ties.  (Writing the arg max from the last slice_update instead of a finish pass was measured: no gain.)"""
    rng = np.random.default_rng(31)
    for (H, W, M, B, iters) in [(41, 57, 21, 3, 5), (16, 16, 2, 1, 1), (30, 33, 29, 2, 3)]:
        rgbs, Us = [], []
        for _ in range(B):
            rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
            U[:, : W // 2] = U[0, : W // 2]  # exact ties between all classes on a strip: the tie rule is exercised
            rgbs.append(rgb)
            Us.append(U)
        cfg = (3, 3, 50, 5, 10, iters)
        q, a_full, _, _ = _gpu_crf(ctx, rgbs, Us, cfg)
        rgb_dev, u_dev = ctx.to_device(np.stack(rgbs)), ctx.to_device(np.stack(Us))
        a_dev = ctx.alloc(B * H * W * 4)
        crf = _lib.Crf(ctx, rgb_dev, B, H, W, cfg[0], cfg[2], cfg[3])
        crf.inference(u_dev, M, cfg[1], cfg[4], iters, None, a_dev)
        a = ctx.to_host(a_dev, (B, H * W), np.int32)
        crf.close()
        assert np.array_equal(a, a_full)
        assert np.array_equal(a, q.argmax(1).astype(np.int32))  # numpy's argmax also takes the first maximum


def test_pixel_major_unary_path_is_bit_identical(ctx):
    """wsc_cam_unary_pm + wsc_crf_inference_pm (unaries written pixel-major by the producer, read in place by the loop)
    against wsc_cam_unary + wsc_crf_inference: same labels and same Q bits, for a class count that needs row padding
    (C + 1 = 21 -> Mp = 24) and one that does not (C + 1 = 8), incl. the n_iters = 0 soft-max."""
    rng = np.random.default_rng(41)
    for (B, C, h, S, iters) in ((2, 20, 9, 45, 3), (1, 7, 5, 32, 0), (1, 7, 5, 32, 2)):
        cam = np.maximum(rng.normal(0.2, 1.0, (B, C, h, h)), 0).astype(np.float32)
        rgbs = np.stack([helpers.synth_crf_case(rng, S, S, 2)[0] for _ in range(B)])
        M, N = C + 1, S * S
        Mp = (M + 3) // 4 * 4
        cam_dev, rgb_dev = ctx.to_device(cam), ctx.to_device(rgbs)
        u_cm, u_pm = ctx.alloc(B * M * N * 4), ctx.alloc(B * N * Mp * 4)
        _lib.cam_unary(ctx, cam_dev, B, C, h, h, S, S, 0.15, u_cm)
        _lib.cam_unary(ctx, cam_dev, B, C, h, h, S, S, 0.15, u_pm, pixel_major=True)
        U = ctx.to_host(u_cm, (B, M, N), np.float32)
        Upm = ctx.to_host(u_pm, (B, N, Mp), np.float32)
        assert np.array_equal(np.transpose(Upm[:, :, :M], (0, 2, 1)), U) and np.all(Upm[:, :, M:] == 0)
        crf = _lib.Crf(ctx, rgb_dev, B, S, S, 1.5, 40.0, 13.0)
        q1, q2 = ctx.alloc(B * M * N * 4), ctx.alloc(B * M * N * 4)
        a1, a2, a3 = ctx.alloc(B * N * 4), ctx.alloc(B * N * 4), ctx.alloc(B * N * 4)
        crf.inference(u_cm, M, 3.0, 10.0, iters, q1, a1)
        crf.inference(u_pm, M, 3.0, 10.0, iters, q2, a2, pixel_major=True)
        crf.inference(u_pm, M, 3.0, 10.0, iters, None, a3, pixel_major=True)  # labels only
        Q1, Q2 = ctx.to_host(q1, (B, M, N), np.float32), ctx.to_host(q2, (B, M, N), np.float32)
        A1, A2, A3 = (ctx.to_host(x, (B, N), np.int32) for x in (a1, a2, a3))
        crf.close()
        if iters == 0:  # the class-major path takes soft-max(-U) in its layout pass (expf, divide), the other in the update kernel
            assert np.abs(Q1 - Q2).max() <= 1e-6 and np.array_equal(A2, A3) and (A1 == A2).mean() >= 0.999
        else:
            assert np.array_equal(Q1, Q2) and np.array_equal(A1, A2) and np.array_equal(A1, A3)
        assert np.array_equal(ctx.to_host(u_pm, (B, N, Mp), np.float32), Upm)  # read in place, never written


def test_crf_gaussian_cache_evicts_least_recently_used():
    """A ctx keeps the Gaussian lattices of 64 image sizes.  The host-built tile vertex sets of the on-chip message path are
    added at a size's SECOND use (a size seen once -- cam_to_ir_label walks hundreds -- never pays the 3-4 ms host pass and
    takes the blur-kernel path: same bits).  A 65th, 66th ... size evicts the least recently used entry no live wsc_crf refers
    to (round 3: later sizes were rebuilt per call, never on chip); an evicted size starts over on its next use and gives
    the same bits as before, and as on a fresh ctx."""
    ctx_a = _lib.Context(0)
    rng = np.random.default_rng(77)
    cfg = (1.5, 3, 20, 10, 10, 2)
    M = 3
    first = last = None
    for i in range(70):
        H, W = 18 + i, 20
        rgb, U, _ = helpers.synth_crf_case(rng, H, W, M)
        q, a, _, _ = _gpu_crf(ctx_a, [rgb], [U], cfg)
        assert not _gpu_crf.on_chip, i      # first use of the size: lattice cached, no vertex sets yet
        q_b, a_b, _, _ = _gpu_crf(ctx_a, [rgb], [U], cfg)
        assert _gpu_crf.on_chip, i          # second use: on chip, identical bits
        assert np.array_equal(q, q_b) and np.array_equal(a, a_b), i
        last = (rgb, U, q, a)
        first = first or last
    q1, a1, _, _ = _gpu_crf(ctx_a, [first[0]], [first[1]], cfg)  # the first size was evicted long ago: starts over
    assert not _gpu_crf.on_chip and np.array_equal(q1, first[2]) and np.array_equal(a1, first[3])
    ctx_b = _lib.Context(0)
    q2, a2, _, _ = _gpu_crf(ctx_b, [last[0]], [last[1]], cfg)
    assert np.array_equal(q2, last[2]) and np.array_equal(a2, last[3])


def test_crf_ragged_batch(ctx):
    """wsc_crf_v: a list of images with their own sizes AND class counts in one object (the reference's per-image loops:
    cam_to_ir_label.py:25-58, 03c_hsn/utilities.py:420-445).  Seven images of three sizes, M from 1 to 21, in mixed order,
    non-contiguous device buffers: Q and labels of every image must be BIT-identical to a wsc_crf call on that image alone
    with its own M (the group loop runs at the group's largest M with the smaller images' missing classes at probability
    zero), and a labels-only ragged call must give the same labels."""
    rng = np.random.default_rng(4711)
    cfg = (1.5, 3, 40, 13, 10, 10)
    specs = [(57, 75, 3), (40, 33, 21), (57, 75, 6), (64, 64, 2), (57, 75, 1), (40, 33, 5), (57, 75, 21)]
    cases = [helpers.synth_crf_case(rng, h, w, m) for (h, w, m) in specs]
    rgb_devs = [ctx.to_device(np.ascontiguousarray(c[0])) for c in cases]
    u_devs = [ctx.to_device(np.ascontiguousarray(c[1])) for c in cases]
    q_devs = [ctx.alloc(m * h * w * 4) for (h, w, m) in specs]
    a_devs = [ctx.alloc(h * w * 4) for (h, w, m) in specs]
    cv = _lib.CrfV(ctx, rgb_devs, [(h, w) for (h, w, m) in specs], cfg[0], cfg[2], cfg[3])
    assert cv.num_groups() == 3
    cv.inference(u_devs, [m for (_, _, m) in specs], cfg[1], cfg[4], cfg[5], q_devs, a_devs)
    qs = [ctx.to_host(q, (m, h * w), np.float32) for q, (h, w, m) in zip(q_devs, specs)]
    labs = [ctx.to_host(a, (h * w,), np.int32) for a, (h, w, m) in zip(a_devs, specs)]
    a2_devs = [ctx.alloc(h * w * 4) for (h, w, m) in specs]
    cv.inference(u_devs, [m for (_, _, m) in specs], cfg[1], cfg[4], cfg[5], None, a2_devs)
    labs2 = [ctx.to_host(a, (h * w,), np.int32) for a, (h, w, m) in zip(a2_devs, specs)]
    cv.close()
    for i, (rgb, U, _) in enumerate(cases):
        q1, a1, _, _ = _gpu_crf(ctx, [rgb], [U], cfg)
        assert np.array_equal(qs[i], q1[0]), (i, specs[i], np.abs(qs[i] - q1[0]).max())
        assert np.array_equal(labs[i], a1[0]) and np.array_equal(labs2[i], a1[0]), (i, specs[i])
