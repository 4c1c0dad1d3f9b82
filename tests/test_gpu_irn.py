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
