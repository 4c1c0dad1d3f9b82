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
from wsscam.net import resnet50_irn, vgg16_irn

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resnet50_irn.npz")
TOL = {_lib.PREC_BF16X3: (2e-4, 2e-3), _lib.PREC_F16: (2e-2, 2e-1)}


@pytest.mark.parametrize("precision", [_lib.PREC_BF16X3, _lib.PREC_F16])
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


@pytest.mark.parametrize("batchnorm", [True, False])
def test_vgg16_irn_vs_oracle(batchnorm):
    sd = irn_ref.make_vgg16_irn_state_dict(seed=2, batchnorm=batchnorm)
    m = vgg16_irn.EdgeDisplacement(None, "voc12" if batchnorm else "adp_morph", "", 20, None, crop_size=96, stride=4,
                                   precision=_lib.PREC_BF16X3)
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
