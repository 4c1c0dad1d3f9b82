"""Generates tests/golden/resnet50_irn.npz by running the REFERENCE module 03b_irn/net/resnet50_irn.Net in this
container (it cannot travel to the GPU box) and checks, in the same process, that oracle/irn_ref.py computes
bit-identical outputs.  Net.__init__ asks torchvision's model zoo for pretrained weights (no network here), so
the backbone factory is replaced by the bare constructor; every parameter then comes from the seeded state dict.

    python oracle/gen_golden_irn.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/03b_irn")

from oracle import cnn_ref, irn_ref  # noqa: E402


def main():
    from net import resnet50 as ref_resnet50

    ref_resnet50.resnet50 = lambda pretrained=True, **kw: ref_resnet50.ResNet(ref_resnet50.Bottleneck, [3, 4, 6, 3], **kw)
    from net import resnet50_irn as ref_irn

    sd = irn_ref.make_resnet50_irn_state_dict(seed=0)
    net = ref_irn.Net()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # the module registers the same tensors under alias names (stage*, backbone.*, edge_layers.*, dp_layers.*)
    # ... and MeanShift twice (mean_shift.* and fc_dp7.4.*): the shared tensors are loaded through their first name
    bad = [k for k in missing if not (k.split(".")[0] in ("stage1", "stage2", "stage3", "stage4", "stage5", "backbone",
                                                          "edge_layers", "dp_layers")
                                      or k == "fc_dp7.4.running_mean" or "num_batches_tracked" in k)]
    assert not bad, bad[:10]
    net.eval()
    rng = np.random.default_rng(11)
    S, h, w, stride = 96, 70, 83, 4
    img = cnn_ref.synth_image(rng, h, w)
    x = torch.from_numpy(cnn_ref.msf_pack(img, (h, w)))  # (2,3,h,w) normalised [orig, flip], no resize
    with torch.no_grad():
        # EdgeDisplacement.forward lines 219-231 around the reference Net.forward (the class itself cannot be
        # constructed: it passes (model_dir, num_classes) to Net.__init__() which takes none)
        fh, fw = (x.size(2) - 1) // stride + 1, (x.size(3) - 1) // stride + 1
        xp = torch.nn.functional.pad(x, [0, S - x.size(3), 0, S - x.size(2)])
        e, d = net(xp)
        e = e[..., :fh, :fw]
        d = d[..., :fh, :fw]
        edge = torch.sigmoid(e[0] / 2 + e[1].flip(-1) / 2)
        dp = d[0]
        e2, d2 = irn_ref.edge_displacement_forward(x, sd, "resnet50", crop_size=S, stride=stride)
    assert torch.equal(edge, e2) and torch.equal(dp, d2), (float((edge - e2).abs().max()), float((dp - d2).abs().max()))
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "resnet50_irn.npz")
    np.savez_compressed(out, x=x.numpy(), edge=edge.numpy(), dp=dp.numpy(), crop_size=S, stride=stride, seed=0)
    print("wrote", out, edge.shape, dp.shape, float(edge.min()), float(edge.max()))


if __name__ == "__main__":
    main()
