"""GPU: IRNet EdgeDisplacement forward (wsc_net_forward_edge) against the torch fp32 oracle -- the ResNet50
flavour on the fixture generated from the reference module, the VGG16 flavour on the restatement.

Tolerances on the sigmoid edge map (values in (0,1)) / the displacement field: bf16x3 2e-4 / 2e-3,
f16 2e-2 / 2e-1 (GroupNorm renormalises every head, so operand rounding does not compound)."""
import os

import numpy as np
import pytest
import torch

from oracle import cnn_ref, irn_ref
from wsscam import _lib
from wsscam.net import m7_irn, resnet50_irn, vgg16_irn

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resnet50_irn.npz")
TOL = {_lib.PREC_F16X3: (1e-4, 1e-3), _lib.PREC_BF16X3: (2e-4, 2e-3), _lib.PREC_F16: (2e-2, 2e-1)}  # F16X3: the package default


@pytest.mark.parametrize("precision", [_lib.PREC_F16X3, _lib.PREC_BF16X3, _lib.PREC_F16])
def test_resnet50_irn_vs_reference_fixture(precision):
    g = np.load(GOLDEN)
    sd = irn_ref.make_resnet50_irn_state_dict(seed=int(g["seed"]))
    m = resnet50_irn.EdgeDisplacement(None, 20, crop_size=int(g["crop_size"]), stride=int(g["stride"]), precision=precision)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    edge, dp = m.forward(g["x"])
    assert edge.shape == g["edge"].shape and dp.shape == g["dp"].shape
    te, td = TOL[precision]
    assert np.abs(edge - g["edge"]).max() <= te, np.abs(edge - g["edge"]).max()
    assert np.abs(dp - g["dp"]).max() <= td * max(1.0, float(np.abs(g["dp"]).max())), np.abs(dp - g["dp"]).max()


@pytest.mark.parametrize("precision", [_lib.PREC_F16X3, _lib.PREC_BF16X3])
@pytest.mark.parametrize("batchnorm", [True, False])
def test_vgg16_irn_vs_oracle(batchnorm, precision):
    sd = irn_ref.make_vgg16_irn_state_dict(seed=2, batchnorm=batchnorm)
    m = vgg16_irn.EdgeDisplacement(None, "voc12" if batchnorm else "adp_morph", "", 20, None, crop_size=96, stride=4,
                                   precision=precision)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    rng = np.random.default_rng(3)
    xs = np.stack([cnn_ref.msf_pack(cnn_ref.synth_image(rng, 77, 90), (77, 90)) for _ in range(2)])
    edge, dp = m.forward_batch(xs)  # a batch of two images
    for b in range(2):
        with torch.no_grad():
            e, d = irn_ref.edge_displacement_forward(torch.from_numpy(xs[b]), sd, "vgg16", crop_size=96, stride=4)
        assert edge[b].shape == tuple(e.shape) == (1, 20, 23) and dp[b].shape == tuple(d.shape)
        assert np.abs(edge[b] - e.numpy()).max() <= 2e-4, np.abs(edge[b] - e.numpy()).max()
        assert np.abs(dp[b] - d.numpy()).max() <= 2e-3 * max(1.0, float(d.abs().max()))


def test_m7_irn_vs_oracle():
    """m7_irn: edge map at 1/2 resolution cropped with the stride-4 feature size (as the reference does), the
    displacement branch at 1/4; fc_dp4 chained on fc_dp3."""
    sd = irn_ref.make_m7_irn_state_dict(seed=4)
    m = m7_irn.EdgeDisplacement(None, "voc12", "", 20, None, crop_size=64, stride=4)  # the package default: f16x3
    m.load_state_dict(sd)
    m.eval().cuda(0)
    rng = np.random.default_rng(6)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 52, 61), (52, 61))
    edge, dp = m.forward(x)
    with torch.no_grad():
        e, d = irn_ref.edge_displacement_forward(torch.from_numpy(x), sd, "m7", crop_size=64, stride=4)
    assert edge.shape == tuple(e.shape) == (1, 13, 16) and dp.shape == tuple(d.shape) == (2, 13, 16)
    assert np.abs(edge - e.numpy()).max() <= 2e-4, np.abs(edge - e.numpy()).max()
    assert np.abs(dp - d.numpy()).max() <= 2e-3 * max(1.0, float(d.abs().max()))


def test_forward_edge_argument_errors(ctx):
    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    net = _lib.Net(ctx, _lib.ARCH_RESNET50_CAM, {k: v.numpy() for k, v in sd.items()}, 20, _lib.PREC_F16)
    x = ctx.to_device(np.zeros((1, 2, 3, 64, 64), np.float32))
    out = ctx.alloc(4 * 16 * 16 * 2)
    with pytest.raises(_lib.WscError):
        net.forward_edge(x, 1, 64, 16, 16, out, out)  # a CAM net has no edge heads
    net.close()
    with pytest.raises(_lib.WscError):  # missing head weights
        _lib.Net(ctx, _lib.ARCH_RESNET50_IRN, {k: v.numpy() for k, v in sd.items()}, 20, _lib.PREC_F16)


@pytest.mark.parametrize("case", [(9, 11, 3, 5, 10, 3), (23, 31, 5, 5, 10, 8), (16, 16, 1, 5, 10, 0), (6, 40, 2, 3, 4, 5)])
def test_propagate_to_edge_vs_dense_oracle(ctx, case):
    """The 2^exp_times stencil applications equal the reference's dense matrix power (upstream irn
    misc/indexing.py restated in oracle/rw_ref.py): same rw maps to fp32 round-off."""
    from oracle import rw_ref
    from wsscam.misc import indexing

    h, w, K, radius, beta, exp_times = case
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.rand(K, h, w, generator=g)
    edge = torch.rand(1, h, w, generator=g) ** 2
    ref = rw_ref.propagate_to_edge(x, edge, radius=radius, beta=beta, exp_times=exp_times).numpy()
    exact = rw_ref.propagate_to_edge(x, edge, radius=radius, beta=beta, exp_times=exp_times, dtype=torch.float64).numpy()
    out = indexing.propagate_to_edge(x.numpy(), edge.numpy(), radius=radius, beta=beta, exp_times=exp_times, ctx=ctx)
    assert out.shape == ref.shape == (K, 1, h, w)
    scale = max(1.0, float(np.abs(exact).max()))
    # against the exact value of the reference's expression: 1e-5; against its fp32 evaluation (whose 8 dense
    # squarings carry more round-off than the 256 stencil steps): 1e-4
    assert np.abs(out - exact).max() <= 1e-5 * scale, np.abs(out - exact).max()
    assert np.abs(out - ref).max() <= 1e-4 * scale, np.abs(out - ref).max()
    out_t = indexing.propagate_to_edge(x, edge, radius=radius, beta=beta, exp_times=exp_times, ctx=ctx)
    assert torch.is_tensor(out_t) and np.array_equal(out_t.numpy(), out)  # bit-reproducible, torch in -> torch out


def test_propagate_batch_equals_single(ctx):
    """Images of different sizes / map counts in one pass give bit-identical maps to one call each."""
    from wsscam.misc import indexing

    rng = np.random.default_rng(2)
    shapes = [(2, 15, 20), (1, 18, 16), (3, 7, 33), (1, 1, 1)]
    xs = [rng.random(s).astype(np.float32) for s in shapes]
    es = [(rng.random((1,) + s[1:]) ** 2).astype(np.float32) for s in shapes]
    outs = indexing.propagate_to_edge_batch(xs, es, beta=10, exp_times=6, ctx=ctx)
    for x, e, o in zip(xs, es, outs):
        single = indexing.propagate_to_edge(x, e, beta=10, exp_times=6, ctx=ctx)
        assert o.shape == single.shape and np.array_equal(o, single)


def test_propagate_tiled_step_is_bit_identical(ctx):
    """rw_step_tile_kernel (16 x 16 pixel tiles, values through LDS, all maps of a pixel per thread) against the flat
    one-thread-per-value kernel (ctx option OPT_RW_TILED = 0): same bits -- maps of 1 ... 7 classes (more than one group of
    four), sizes that are not multiples of the tile, a one-pixel image."""
    from wsscam.misc import indexing

    rng = np.random.default_rng(5)
    shapes = [(2, 94, 125), (7, 33, 47), (1, 16, 16), (5, 1, 1), (3, 17, 5)]
    xs = [rng.random(s).astype(np.float32) for s in shapes]
    es = [(rng.random((1,) + s[1:]) ** 2).astype(np.float32) for s in shapes]
    with ctx.option(_lib.OPT_RW_TILED, 0):
        ref = indexing.propagate_to_edge_batch(xs, es, beta=10, exp_times=5, ctx=ctx)
    with ctx.option(_lib.OPT_RW_TILED, 1):
        out = indexing.propagate_to_edge_batch(xs, es, beta=10, exp_times=5, ctx=ctx)
    for r, o in zip(ref, out):
        assert r.shape == o.shape and np.array_equal(r, o)


def test_config4_real_size_walk_and_edge_net(ctx):
    """BASELINE config 4 at its REAL size: a 375 x 500 VOC image -- the random walk of K = 2 strided CAMs on
    94 x 125 (exp_times 8 = 256 stencil steps, beta 10) through wsc_rw_propagate_batch's TILED path (32 images per pass: the
    batch fills the chip with 16 x 16 tiles, csrc/rw.hip) against the float64 sparse statement of the reference's dense
    matrix-power form (oracle/rw_ref.py::propagate_to_edge_sparse; the 11 750^2 matrix itself is never built), and
    wsc_net_forward_edge on the [orig, flip] pair zero-padded to crop 512 in the headline mode f16x3 against oracle/irn_ref.py
    (03b_irn/step/make_sem_seg_labels.py:46-110, net/resnet50_irn.py:210-232)."""
    from oracle import rw_ref
    from wsscam.misc import indexing

    rng = np.random.default_rng(31)
    K, h, w, n_img = 2, 94, 125, 32
    xs = [rng.random((K, h, w)).astype(np.float32) for _ in range(n_img)]
    es = [(rng.random((1, h, w)) ** 2).astype(np.float32) for _ in range(n_img)]
    with ctx.option(_lib.OPT_RW_TILED, 1):  # (the default picks it for this batch as well; forced so the test says what it covers)
        outs = indexing.propagate_to_edge_batch(xs, es, beta=10, exp_times=8, ctx=ctx)
    worst = 0.0
    for i in (0, 13, 31):
        exact = rw_ref.propagate_to_edge_sparse(xs[i], es[i], radius=5, beta=10, exp_times=8)
        assert outs[i].shape == exact.shape == (K, 1, h, w)
        scale = max(1.0, float(np.abs(exact).max()))
        err = float(np.abs(outs[i] - exact).max()) / scale
        worst = max(worst, err)
        assert err <= 1e-5, (i, err)  # the bound of the small-grid test against the exact value of the reference's expression
        assert float(np.abs(exact).max()) > 1e-3  # (the walk has not collapsed to zero: the comparison is not vacuous)
    print("config 4 walk at 94 x 125, 256 steps, tiled path: max |device - float64 oracle| = %.2e x scale" % worst)

    sd = irn_ref.make_resnet50_irn_state_dict(seed=3)
    m = resnet50_irn.EdgeDisplacement(None, 20, crop_size=512, stride=4, precision=_lib.PREC_F16X3)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 375, 500), None)  # (2, 3, 375, 500): make_sem_seg_labels feeds the native size
    assert x.shape == (2, 3, 375, 500)
    edge, dp = m.forward(x)
    with torch.no_grad():
        e, d = irn_ref.edge_displacement_forward(torch.from_numpy(x), sd, "resnet50", crop_size=512, stride=4)
    assert edge.shape == tuple(e.shape) == (1, h, w) and dp.shape == tuple(d.shape) == (2, h, w)
    te, td = TOL[_lib.PREC_F16X3]
    ee, ed = float(np.abs(edge - e.numpy()).max()), float(np.abs(dp - d.numpy()).max()) / max(1.0, float(d.abs().max()))
    print("config 4 edge net at crop 512 (f16x3): edge %.2e, displacement %.2e" % (ee, ed))
    assert ee <= te and ed <= td, (ee, ed)


def test_path_index_tables():
    """PathIndex(radius=5): 34 directions in the upper half plane, paths include both end points, destinations
    first; radius 10 (cam_to_ir_label / train_irn) also builds."""
    from wsscam.misc.indexing import PathIndex

    pi = PathIndex(5, default_size=(12, 20))
    dirs, start, yx = pi.device_tables()
    assert dirs.shape == (34, 2) and start[0] == 0 and start[-1] == len(yx)
    assert all(d[0] > 0 or (d[0] == 0 and d[1] > 0) for d in dirs.tolist())
    for d in range(34):
        path = yx[start[d]:start[d + 1]].tolist()
        assert path[0] == dirs[d].tolist() and [0, 0] in path
    assert pi.search_dst.shape == (34, 2) and pi.radius_floor == 4
    assert sum(p.shape[0] for p in pi.path_indices) == 34 and pi.src_indices.shape[0] == (12 - 4) * (20 - 8)
    assert PathIndex(10).device_tables()[0].shape[0] > 100


def test_make_sem_seg_labels_end_to_end(tmp_path):
    """make_cam.run -> make_sem_seg_labels.run on the files it wrote (VOC flavour): label PNGs against the oracle
    chain irn_ref (network) -> rw_ref (dense random walk, as the reference computes it) -> torch interpolate."""
    import types

    from PIL import Image

    from oracle import rw_ref
    from wsscam.step import make_cam, make_sem_seg_labels

    rng = np.random.default_rng(9)
    S = 65
    sizes = [(60, 80), (72, 64), (50, 50)]
    labels = [np.zeros(20, np.float32) for _ in sizes]
    labels[0][[2, 5]] = 1
    labels[1][[11]] = 1
    # labels[2] empty: make_cam writes empty arrays and the label map is all background
    data = [{"name": "2008_%06d" % i, "img": cnn_ref.msf_pack(cnn_ref.synth_image(rng, *sz), (S, S)), "size": sz,
             "label": lb} for i, (sz, lb) in enumerate(zip(sizes, labels))]
    cam_dir, seg_dir, clr_dir = tmp_path / "cam", tmp_path / "seg", tmp_path / "clr"
    cam_sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=0)
    a1 = types.SimpleNamespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                               use_cls=None, model_id="resnet50", cam_weights_name=None, state_dict=cam_sd,
                               dataset_obj=data, split="train_aug", cam_out_dir=str(cam_dir), n_gpus=1,
                               cam_batch_images=4, cam_precision=_lib.PREC_F16X3)
    make_cam.run(a1)
    irn_sd = irn_ref.make_vgg16_irn_state_dict(seed=5)
    colours = {"bg": np.array([(0, 0, 0)]), "fg": np.array([(8 * i + 8, 255 - 8 * i, 17 * (i % 15)) for i in range(20)])}
    a2 = types.SimpleNamespace(irn_network="net.vgg16_irn", model_dir=None, dataset="voc12", tag="", num_classes=20,
                               use_cls=None, irn_weights_name=None, state_dict=irn_sd, dataset_obj=data, split="train_aug",
                               cam_out_dir=str(cam_dir), sem_seg_out_dir=str(seg_dir), sem_seg_clr_out_dir=str(clr_dir),
                               beta=10, exp_times=8, sem_seg_bg_thres=0.25, class_colours=colours, overlay_r=0.75,
                               n_gpus=1, irn_crop_size=96, irn_precision=_lib.PREC_F16X3)
    make_sem_seg_labels.run(a2)
    for d in data:
        png = np.asarray(Image.open(seg_dir / (d["name"] + ".png")))
        assert png.shape == d["size"] and png.dtype == np.uint8
        assert (clr_dir / (d["name"] + ".png")).exists()
        cam = np.load(cam_dir / (d["name"] + ".npy"), allow_pickle=True).item()
        if len(cam["keys"]) == 0:
            assert not png.any()
            continue
        with torch.no_grad():
            edge, _ = irn_ref.edge_displacement_forward(torch.from_numpy(d["img"]), irn_sd, "vgg16", crop_size=96, stride=4)
            cams = torch.from_numpy(cam["cam"])
            if edge.shape[1:] != cams.shape[1:]:
                edge = torch.nn.functional.interpolate(edge.unsqueeze(0), size=cams.shape[1:], mode="bilinear",
                                                       align_corners=False)[0]
            rw = rw_ref.propagate_to_edge(cams, edge, beta=10, exp_times=8, radius=5)
            up = torch.nn.functional.interpolate(rw, size=d["size"], mode="bilinear", align_corners=False)[..., 0, :d["size"][0], :d["size"][1]]
            up = up / torch.max(up)
            up_bg = torch.nn.functional.pad(up, (0, 0, 0, 0, 1, 0), value=0.25)
            ref = np.pad(cam["keys"] + 1, (1, 0), mode="constant")[torch.argmax(up_bg, dim=0).numpy()]
        assert set(np.unique(png)) <= set(np.pad(cam["keys"] + 1, (1, 0)).tolist())
        assert (png == ref).mean() >= 0.99, (png == ref).mean()


def test_sem_seg_batches_in_flight_equal_serial():
    """Round 6: make_sem_seg_labels keeps several batches in flight on their own streams (sem_seg_batches); the label maps of
    a batch do not depend on its lane: seven images in batches of two on three lanes == sem_seg_batch one batch after the
    other on the model's own context, bit for bit, in batch order."""
    import types

    from wsscam.net import vgg16_irn
    from wsscam.step import make_sem_seg_labels as mssl

    rng = np.random.default_rng(31)
    sd = irn_ref.make_vgg16_irn_state_dict(seed=6)
    model = vgg16_irn.EdgeDisplacement(None, "voc12", "", 20, None, crop_size=96, stride=4, precision=_lib.PREC_F16X3)
    model.load_state_dict(sd, strict=False)
    model.eval().cuda(0)
    packs, cams = [], []
    for i in range(7):
        packs.append({"name": "i%d" % i, "img": rng.normal(0, 1, (2, 3, 72, 88)).astype(np.float32), "size": (72, 88)})
        K = 1 + i % 3
        cams.append({"keys": np.sort(rng.choice(20, K, replace=False)), "cam": rng.random((K, 18, 22)).astype(np.float32)})
    args = types.SimpleNamespace(dataset="voc12", beta=10, exp_times=5, sem_seg_bg_thres=0.25)
    batches = [(packs[i:i + 2], cams[i:i + 2]) for i in range(0, 7, 2)]
    serial = [mssl.sem_seg_batch(model, p, c, args) for p, c in batches]
    for chain in (False, True):  # (chain_stacks: the lanes' network passes take turns on the device, _lib.StackChain)
        lanes = mssl.sem_seg_batches(model, batches, args, n_lanes=3, chain_stacks=chain)
        assert len(lanes) == len(serial)
        for a, b in zip(lanes, serial):
            assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    # the sink form (what _work uses): every batch delivered once, on its lane's thread
    got = {}
    mssl.sem_seg_batches(model, [lambda bt=bt: bt for bt in batches], args, n_lanes=2,
                         sink=lambda bi, pk, preds: got.__setitem__(bi, (len(pk), [np.asarray(x).copy() for x in preds])))
    assert sorted(got) == list(range(len(batches)))
    for bi, (n, preds) in got.items():
        assert n == len(batches[bi][0]) and all(np.array_equal(x, y) for x, y in zip(preds, serial[bi]))


@pytest.mark.parametrize("dataset", ["adp_func", "deepglobe"])
def test_sem_seg_other_datasets(dataset):
    """The ADP and DeepGlobe branches of make_sem_seg_labels._work (:71-99): no background padding, keys used as
    they are; DeepGlobe walks on CAMs downsampled by 6 and writes a quarter-size label map."""
    import types

    from oracle import rw_ref
    from wsscam.step import make_sem_seg_labels as mssl

    sd = irn_ref.make_vgg16_irn_state_dict(seed=7, batchnorm=dataset != "adp_func")
    m = vgg16_irn.EdgeDisplacement(None, dataset, "", 5, None, crop_size=96, stride=4, precision=_lib.PREC_F16X3)
    m.load_state_dict(sd)
    m.eval().cuda(0)
    rng = np.random.default_rng(12)
    size = (68, 72) if dataset == "adp_func" else (288, 336)
    x = cnn_ref.msf_pack(cnn_ref.synth_image(rng, 65, 65), (65, 65))
    cam_hw = ((size[0] - 1) // 4 + 1, (size[1] - 1) // 4 + 1)
    cam = {"keys": np.array([0, 1, 3]), "cam": rng.random((3,) + cam_hw).astype(np.float32)}
    args = types.SimpleNamespace(dataset=dataset, beta=10, exp_times=6, sem_seg_bg_thres=0.25)
    pred = mssl.sem_seg_one(m, {"img": x, "size": size, "name": "t"}, cam, args)
    with torch.no_grad():
        edge, _ = irn_ref.edge_displacement_forward(torch.from_numpy(x), sd, "vgg16", crop_size=96, stride=4)
        cams = torch.from_numpy(cam["cam"])
        out_size = size
        if dataset == "deepglobe":
            cams = torch.nn.functional.interpolate(cams.unsqueeze(0), size=[v // 6 for v in cams.shape[1:]], mode="bilinear",
                                                   align_corners=False)[0]
            out_size = (size[0] // 4, size[1] // 4)
        edge = torch.nn.functional.interpolate(edge.unsqueeze(0), size=cams.shape[1:], mode="bilinear", align_corners=False)[0]
        rw = rw_ref.propagate_to_edge(cams, edge, beta=10, exp_times=6, radius=5)
        up = torch.nn.functional.interpolate(rw, size=out_size, mode="bilinear", align_corners=False)[..., 0, :out_size[0], :out_size[1]]
        ref = cam["keys"][torch.argmax(up / torch.max(up), dim=0).numpy()]
    assert pred.shape == ref.shape == out_size
    assert (pred == ref).mean() >= 0.99, (pred == ref).mean()
