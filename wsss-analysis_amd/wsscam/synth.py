"""Synthetic workload of SURVEY.md section 8(d): VOC-like images, seeded random weights of the three CAM
networks.  There is no dataset and no checkpoint offline, so bench.py and the drivers' dry runs use these;
numpy only (no torch, no oracle/): product-side code.

The weights are Kaiming-scaled so activations stay O(1) through 50 layers; BatchNorm buffers are
randomised to non-trivial values so folding bugs show.  State-dict keys are the reference's
(`resnet50.*`, `classifier.weight`: 03b_irn/net/resnet50_cam.py:12-20; `<root>.<layer>.<idx>.*`,
`<root>.classifier.0.*`: net/vgg16.py:11-24, net/m7.py:11-23)."""
import math

import numpy as np

from .net.common import PLAIN_CFG
from .voc12.dataloader import TorchvisionNormalize, msf_pack, resize_bilinear_f64  # noqa: F401  (re-exported)

RESNET_BLOCKS = (3, 4, 6, 3)
RESNET_PLANES = (64, 128, 256, 512)
RESNET_CAM_STRIDES = (2, 2, 2, 1)  # 03b_irn/net/resnet50_cam.py:15
# native sizes of the synthetic set: 60 % 375x500, 20 % 500x375, 10 % 333x500, 10 % 500x500
VOC_SIZES = [(375, 500)] * 6 + [(500, 375)] * 2 + [(333, 500)] + [(500, 500)]


def synth_image(rng, H, W):
    """uint8 RGB: six soft-edged ellipses of random colour over a low-frequency gradient + N(0, 8) noise
    (non-degenerate bilateral lattices)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), np.float32)
    base = rng.uniform(40, 200, 3)
    grad = rng.uniform(-60, 60, (2, 3))
    img += base + (yy / H)[..., None] * grad[0] + (xx / W)[..., None] * grad[1]
    for _ in range(6):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ry, rx = rng.uniform(0.08, 0.35) * H, rng.uniform(0.08, 0.35) * W
        col = rng.uniform(0, 255, 3)
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        a = 1.0 / (1.0 + np.exp(np.minimum((d - 1.0) * 6.0, 60.0)))
        img = img * (1 - a[..., None]) + col * a[..., None]
    img += rng.normal(0, 8, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _bn(rng, c, prefix, sd):
    sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    sd[prefix + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)


def _conv(rng, cout, cin, k, gain=1.0):
    return (rng.standard_normal((cout, cin, k, k)) * (gain * math.sqrt(2.0 / (cin * k * k)))).astype(np.float32)


def resnet50_cam_state_dict(num_classes=20, seed=0):
    rng = np.random.default_rng(1000 + seed)
    sd = {"resnet50.conv1.weight": _conv(rng, 64, 3, 7)}
    _bn(rng, 64, "resnet50.bn1", sd)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(zip(RESNET_PLANES, RESNET_BLOCKS, (1,) + RESNET_CAM_STRIDES[1:])):
        for bi in range(blocks):
            pre = "resnet50.layer%d.%d" % (li + 1, bi)
            s = stride if bi == 0 else 1
            sd[pre + ".conv1.weight"] = _conv(rng, planes, inplanes, 1)
            _bn(rng, planes, pre + ".bn1", sd)
            sd[pre + ".conv2.weight"] = _conv(rng, planes, planes, 3)
            _bn(rng, planes, pre + ".bn2", sd)
            sd[pre + ".conv3.weight"] = _conv(rng, planes * 4, planes, 1, gain=0.5)  # small residual branch
            _bn(rng, planes * 4, pre + ".bn3", sd)
            if bi == 0 and (s != 1 or inplanes != planes * 4):
                sd[pre + ".downsample.0.weight"] = _conv(rng, planes * 4, inplanes, 1, gain=0.7)
                _bn(rng, planes * 4, pre + ".downsample.1", sd)
            inplanes = planes * 4
    sd["classifier.weight"] = (rng.standard_normal((num_classes, 2048, 1, 1)) * 0.01).astype(np.float32)
    return sd


def plain_state_dict(root, num_classes=20, batchnorm=True, seed=0):
    """vgg16 / m7 stack (common_cnn.make_layers: conv(bias) -> ReLU -> BatchNorm) + Linear classifier."""
    rng = np.random.default_rng(2000 + seed)
    sd = {}
    cin = 3
    for lname, layer in PLAIN_CFG[root]:
        idx = 0
        for v in layer:
            if v in ("M", "D"):
                idx += 1
                continue
            key = "%s.%s.%d" % (root, lname, idx)
            sd[key + ".weight"] = _conv(rng, v, cin, 3)
            sd[key + ".bias"] = (rng.standard_normal(v) * 0.05).astype(np.float32)
            if batchnorm:
                _bn(rng, v, "%s.%s.%d" % (root, lname, idx + 2), sd)
                idx += 3
            else:
                idx += 2
            cin = v
    sd[root + ".classifier.0.weight"] = (rng.standard_normal((num_classes, cin)) * 0.05).astype(np.float32)
    sd[root + ".classifier.0.bias"] = (rng.standard_normal(num_classes) * 0.05).astype(np.float32)
    return sd


# IRNet EdgeDisplacement heads (03b_irn/net/resnet50_irn.py:24-77, vgg16_irn.py:30-98 with ds_fac = 0.25):
# (name, backbone stage feeding it, output channels)
IRN_HEADS = {
    "resnet50": {"stage_channels": (64, 256, 512, 1024, 2048),
                 "heads": [("fc_edge1", 1, 32), ("fc_edge2", 2, 32), ("fc_edge3", 3, 32), ("fc_edge4", 4, 32), ("fc_edge5", 5, 32),
                           ("fc_dp1", 1, 64), ("fc_dp2", 2, 128), ("fc_dp3", 3, 256), ("fc_dp4", 4, 256), ("fc_dp5", 5, 256)]},
    "vgg16": {"stage_channels": (64, 128, 256, 512, 1024),
              "heads": [("fc_edge1", 1, 32), ("fc_edge2", 2, 32), ("fc_edge3", 3, 32), ("fc_edge4", 4, 32), ("fc_edge5", 5, 32),
                        ("fc_dp1", 1, 64), ("fc_dp2", 2, 128), ("fc_dp3", 3, 256), ("fc_dp4", 4, 256), ("fc_dp5", 5, 256)]},
}


def irn_state_dict(arch="resnet50", seed=0):
    """Seeded random weights of an IRNet EdgeDisplacement network (there are no trained IRNet weights offline): the CAM
    backbone's state dict without its classifier + the Conv1x1 / GroupNorm heads of both branches, the final edge conv
    (with bias), the final displacement conv and the mean-shift buffer.  Keys as the reference's state dict has them."""
    rng = np.random.default_rng(3000 + seed)
    if arch == "resnet50":
        sd = resnet50_cam_state_dict(20, seed=seed)
        del sd["classifier.weight"]
    else:
        sd = plain_state_dict("vgg16", 20, True, seed=seed)
    cs = IRN_HEADS[arch]["stage_channels"]

    def head(name, cin, cout):
        sd[name + ".0.weight"] = _conv(rng, cout, cin, 1)
        sd[name + ".1.weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sd[name + ".1.bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)

    for name, src, cout in IRN_HEADS[arch]["heads"]:
        head(name, cs[src - 1], cout)
    head("fc_dp6", 768, 256)
    head("fc_dp7", 448, 256)
    sd["fc_edge6.weight"] = (rng.standard_normal((1, 160, 1, 1)) * 0.1).astype(np.float32)
    sd["fc_edge6.bias"] = (rng.standard_normal(1) * 0.1).astype(np.float32)
    sd["fc_dp7.3.weight"] = _conv(rng, 2, 256, 1)
    sd["mean_shift.running_mean"] = (rng.standard_normal(2) * 0.1).astype(np.float32)
    return sd


def image_batch(batch, S, seed=0, with_native=False):
    """-> (x float32 (B,2,3,S,S) MSF items, rgb uint8 (B,S,S,3) resized images, native sizes [(H0,W0)]
    [, the native-size uint8 images])."""
    rng = np.random.default_rng(20121 + seed)
    norm = TorchvisionNormalize("int")
    xs, rgbs, sizes, native = [], [], [], []
    for i in range(batch):
        H0, W0 = VOC_SIZES[(i + seed) % len(VOC_SIZES)]
        img = synth_image(rng, H0, W0)
        native.append(img)
        r = resize_bilinear_f64(img, (S, S))
        x = np.transpose(norm(r), (2, 0, 1))
        xs.append(np.stack([x, np.flip(x, -1)], 0))
        rgbs.append(np.clip(np.rint(r), 0, 255).astype(np.uint8))
        sizes.append((H0, W0))
    out = (np.ascontiguousarray(np.stack(xs), dtype=np.float32), np.ascontiguousarray(np.stack(rgbs)), sizes)
    return out + (native,) if with_native else out


def adp_image(rng, H, W):
    """ADP-like histology patch: pink / purple blobs on a near-white slide background, so the
    0.75 * expit(4 * (mean_rgb - 240)) background activation of modify_by_htt is exercised."""
    img = np.full((H, W, 3), 244.0) + rng.normal(0, 3, (H, W, 3))
    yy, xx = np.mgrid[:H, :W]
    for _ in range(max(4, H * W // 12000)):
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(0.04, 0.18) * min(H, W)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        img[m] = rng.uniform([150, 60, 130], [220, 130, 200]) + rng.normal(0, 6, (int(m.sum()), 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def specialise_classifier(sd, root, num_classes, features, seed=0, amplitude=0.5):
    """Replaces `<root>.classifier.0.weight` (C x F) by a head calibrated to the (random) features: every class responds to
    the MEAN feature vector m of a sample batch with the same logit 1, plus a class-specific random direction orthogonal to
    m (whitened per channel, `amplitude` logit units of standard deviation over the sample's pixels).
    Why: with i.i.d. weights the few classes whose alpha has the largest component along m -- the features are post-ReLU,
    so m is large -- have the largest Grad-CAM map at EVERY pixel, and a Grad-CAM -> arg-max -> CRF chain degenerates to 2-4
    classes per image (1 for the functional types).  A trained classifier's classes win in different places; exchangeable
    class maps reproduce that with random weights: 10-15 morphological and 3 functional classes carry mass per ADP-like patch
    (HSN bench, BASELINE config 5: the dense CRFs then run at a realistic M).
    features: float array (..., F) of final feature maps over a sample batch."""
    rng = np.random.default_rng(977 + seed)
    W = np.asarray(sd[root + ".classifier.0.weight"])
    C, F = W.shape
    assert C == num_classes
    feats = np.asarray(features, dtype=np.float64).reshape(-1, F)
    m, s_f = feats.mean(0), np.maximum(feats.std(0), 1e-6)
    mn = max(float(np.linalg.norm(m)), 1e-12)
    mh = m / mn
    Wn = np.zeros((C, F))
    for c in range(C):
        g = rng.standard_normal(F) / s_f / np.sqrt(F)
        g -= (g @ mh) * mh
        Wn[c] = mh / mn + amplitude * g
    sd[root + ".classifier.0.weight"] = Wn.astype(np.float32)
    return sd
