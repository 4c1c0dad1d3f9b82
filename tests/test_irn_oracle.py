"""CPU: oracle/irn_ref.py reproduces the fixture generated from the reference's own resnet50_irn.Net module
(oracle/gen_golden_irn.py asserts bit-equality in-process; across processes oneDNN may reorder fp32 sums)."""
import os

import numpy as np
import torch

from oracle import irn_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resnet50_irn.npz")


def test_resnet50_irn_matches_reference_module():
    g = np.load(GOLDEN)
    sd = irn_ref.make_resnet50_irn_state_dict(seed=int(g["seed"]))
    torch.set_num_threads(8)
    with torch.no_grad():
        edge, dp = irn_ref.edge_displacement_forward(torch.from_numpy(g["x"]), sd, "resnet50",
                                                     crop_size=int(g["crop_size"]), stride=int(g["stride"]))
    assert edge.shape == g["edge"].shape and dp.shape == g["dp"].shape
    assert np.allclose(edge.numpy(), g["edge"], atol=2e-5) and np.allclose(dp.numpy(), g["dp"], atol=2e-4)


def test_vgg16_irn_restatement_shapes():
    """ds_fac = 0.25: stage1 at 1/2 resolution + stride-2 heads, everything ends at 1/4 resolution."""
    sd = irn_ref.make_vgg16_irn_state_dict(seed=1)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        e, d = irn_ref.irn_net_forward(x, sd, "vgg16")
        edge, dp = irn_ref.edge_displacement_forward(x[..., :50, :61], sd, "vgg16", crop_size=64, stride=4)
    assert e.shape == (2, 1, 16, 16) and d.shape == (2, 2, 16, 16)
    assert edge.shape == (1, 13, 16) and dp.shape == (2, 13, 16)
    assert float(edge.min()) > 0 and float(edge.max()) < 1


def test_sparse_random_walk_oracle_equals_dense_form():
    """oracle/rw_ref.py::propagate_to_edge_sparse (what the VOC-size GPU test checks the tiled random-walk path against)
    is the dense matrix-power form of the same file, evaluated in float64, on grids the dense form still fits."""
    from oracle import rw_ref

    for (K, h, w, radius, beta, exp_times) in ((2, 13, 17, 5, 10, 4), (3, 9, 22, 5, 8, 6), (1, 6, 7, 3, 10, 3)):
        g = torch.Generator().manual_seed(h * 100 + w)
        x = torch.rand(K, h, w, generator=g)
        edge = torch.rand(1, h, w, generator=g) ** 2
        dense = rw_ref.propagate_to_edge(x, edge, radius=radius, beta=beta, exp_times=exp_times, dtype=torch.float64).numpy()
        sparse = rw_ref.propagate_to_edge_sparse(x, edge, radius=radius, beta=beta, exp_times=exp_times)
        assert sparse.shape == dense.shape and sparse.dtype == np.float64
        assert np.abs(sparse - dense).max() <= 1e-12 * max(1.0, float(np.abs(dense).max())), np.abs(sparse - dense).max()
