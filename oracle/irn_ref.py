"""TEST INFRASTRUCTURE ONLY (oracle) -- torch-CPU fp32 restatement of the IRNet EdgeDisplacement networks.

Only tests/ (and fixture generators under oracle/) may import this.  Follows
  03b_irn/net/resnet50_irn.py:8-132 (Net), :210-232 (EdgeDisplacement.forward)
  03b_irn/net/vgg16_irn.py:8-212 (Net, ds_fac = 0.25), :301-321 (EdgeDisplacement.forward)
The ResNet50 flavour is pinned against the reference's own module (imported by oracle/gen_golden_irn.py in the
build container; tests/golden/resnet50_irn.npz); the VGG16 flavour cannot be imported (its backbone
constructor loads weight files that are not in the tree) and is a restatement of the cited lines.
"""
import torch
import torch.nn.functional as F

from oracle import cnn_ref

# (name, source, conv stride, out channels, GroupNorm groups, upsample factor)
RESNET50_HEADS = {
    "edge": [("fc_edge1", 1, 1, 32, 4, 1), ("fc_edge2", 2, 1, 32, 4, 1), ("fc_edge3", 3, 1, 32, 4, 2),
             ("fc_edge4", 4, 1, 32, 4, 4), ("fc_edge5", 5, 1, 32, 4, 4)],
    "dp": [("fc_dp1", 1, 1, 64, 8, 1), ("fc_dp2", 2, 1, 128, 16, 1), ("fc_dp3", 3, 1, 256, 16, 1),
           ("fc_dp4", 4, 1, 256, 16, 2), ("fc_dp5", 5, 1, 256, 16, 2)],
    "stage_channels": (64, 256, 512, 1024, 2048),
}
VGG16_HEADS = {  # ds_fac == 0.25 branch, vgg16_irn.py:30-98
    "edge": [("fc_edge1", 1, 2, 32, 4, 1), ("fc_edge2", 2, 1, 32, 4, 1), ("fc_edge3", 3, 1, 32, 4, 2),
             ("fc_edge4", 4, 1, 32, 4, 2), ("fc_edge5", 5, 1, 32, 4, 2)],
    "dp": [("fc_dp1", 1, 2, 64, 8, 1), ("fc_dp2", 2, 1, 128, 16, 1), ("fc_dp3", 3, 1, 256, 16, 1),
           ("fc_dp4", 4, 1, 256, 16, 1), ("fc_dp5", 5, 1, 256, 16, 1)],
    "stage_channels": (64, 128, 256, 512, 1024),
}


M7_HEADS = {  # m7_irn.py:24-69: (name, source, conv stride, out channels, groups, upsample); fc_dp4 reads fc_dp3's output
    "edge": [("fc_edge1", 1, 1, 32, 4, 1), ("fc_edge2", 2, 1, 32, 4, 2), ("fc_edge3", 3, 1, 32, 4, 4)],
    "dp": [("fc_dp1", 1, 2, 64, 8, 1), ("fc_dp2", 2, 1, 128, 16, 1), ("fc_dp3", 3, 1, 256, 16, 1)],
    "stage_channels": (64, 128, 256),
}
M7_STAGES = [[("layer1", [64, 64, "M"])], [("layer2", [128, 128, "M"])],
             [("layer3_p1", [256, 256, 256]), ("layer3_p2", ["M", "D"])]]  # m7_irn.py:19-21


def make_m7_irn_state_dict(seed=0):
    sd = cnn_ref.make_plain_state_dict("m7", cnn_ref.M7_CFG, 20, True, seed=seed)
    g = torch.Generator().manual_seed(seed + 1000)

    def head(name, cin, cout):
        sd[name + ".0.weight"] = cnn_ref._conv_w(g, cout, cin, 1)
        sd[name + ".1.weight"] = torch.empty(cout).uniform_(0.5, 1.5, generator=g)
        sd[name + ".1.bias"] = torch.randn(cout, generator=g) * 0.1

    for name, src, _, cout, _, _ in M7_HEADS["edge"] + M7_HEADS["dp"]:
        head(name, M7_HEADS["stage_channels"][src - 1], cout)
    head("fc_dp4", 256, 256)
    head("fc_dp5", 448, 256)
    sd["fc_edge4.weight"] = torch.randn(1, 96, 1, 1, generator=g) * 0.1
    sd["fc_edge4.bias"] = torch.randn(1, generator=g) * 0.1
    sd["fc_dp5.3.weight"] = cnn_ref._conv_w(g, 2, 256, 1)
    sd["mean_shift.running_mean"] = torch.randn(2, generator=g) * 0.1
    return sd


def m7_net_forward(x, sd):
    """m7_irn.Net.forward (:96-112)."""
    xs = []
    for stage in M7_STAGES:
        x = cnn_ref.plain_features(x, sd, "m7", stage)
        xs.append(x)
    e = [_head(xs[src - 1], sd, n, st, g, up) for n, src, st, _, g, up in M7_HEADS["edge"]]
    h1, w1 = e[0].shape[2], e[0].shape[3]
    edge_out = F.conv2d(torch.cat([e[0], e[1][..., :h1, :w1], e[2][..., :h1, :w1]], dim=1), sd["fc_edge4.weight"],
                        sd["fc_edge4.bias"])
    d = [_head(xs[src - 1], sd, n, st, g, up) for n, src, st, _, g, up in M7_HEADS["dp"]]
    dp4 = _head(d[2], sd, "fc_dp4", 1, 16, 2)
    hid = _head(torch.cat([d[0], d[1], dp4], dim=1), sd, "fc_dp5", 1, 16, 1)
    dp_out = F.conv2d(hid, sd["fc_dp5.3.weight"]) - sd["mean_shift.running_mean"].view(1, 2, 1, 1)
    return edge_out, dp_out


def add_head_weights(sd, heads, seed):
    """Random weights for the two branches (there are no trained IRNet weights offline)."""
    g = torch.Generator().manual_seed(seed)
    cs = heads["stage_channels"]

    def head(name, cin, cout):
        sd[name + ".0.weight"] = cnn_ref._conv_w(g, cout, cin, 1)
        sd[name + ".1.weight"] = torch.empty(cout).uniform_(0.5, 1.5, generator=g)
        sd[name + ".1.bias"] = torch.randn(cout, generator=g) * 0.1

    for name, src, _, cout, _, _ in heads["edge"] + heads["dp"]:
        head(name, cs[src - 1], cout)
    head("fc_dp6", 768, 256)
    head("fc_dp7", 448, 256)
    sd["fc_edge6.weight"] = torch.randn(1, 160, 1, 1, generator=g) * 0.1
    sd["fc_edge6.bias"] = torch.randn(1, generator=g) * 0.1
    sd["fc_dp7.3.weight"] = cnn_ref._conv_w(g, 2, 256, 1)
    sd["mean_shift.running_mean"] = torch.randn(2, generator=g) * 0.1
    return sd


def make_resnet50_irn_state_dict(seed=0):
    sd = cnn_ref.make_resnet50_cam_state_dict(20, seed=seed)
    del sd["classifier.weight"]
    return add_head_weights(sd, RESNET50_HEADS, seed + 1000)


def make_vgg16_irn_state_dict(seed=0, batchnorm=True):
    sd = cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, 20, batchnorm, seed=seed)
    return add_head_weights(sd, VGG16_HEADS, seed + 1000)


def resnet50_stages(x, sd):
    """x1..x5 of resnet50_irn.Net.forward (:111-115): stage1 = conv1+bn1+relu+maxpool, stage k = layer k-1."""
    x = F.conv2d(x, sd["resnet50.conv1.weight"], stride=2, padding=3)
    x = F.relu(cnn_ref._fixed_bn(x, sd, "resnet50.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    out = [x]
    for li, (blocks, stride) in enumerate(zip(cnn_ref.RESNET_BLOCKS, (1, 2, 2, 1))):
        for bi in range(blocks):
            x = cnn_ref._bottleneck(x, sd, "resnet50.layer%d.%d" % (li + 1, bi), stride if bi == 0 else 1)
        out.append(x)
    return out


def vgg16_stages(x, sd):
    """x1..x5 of vgg16_irn.Net.forward (:193-197): stage k = vgg16.layer k (common_cnn.make_layers)."""
    out = []
    for lname, layer in cnn_ref.VGG16_CFG:
        x = cnn_ref.plain_features(x, sd, "vgg16", [(lname, layer)])
        out.append(x)
    return out


def _head(x, sd, name, stride, groups, up, relu=True):
    # nn.Sequential(Conv2d(1x1, bias=False[, stride]), GroupNorm, [Upsample(bilinear, align_corners=False)], ReLU)
    x = F.conv2d(x, sd[name + ".0.weight"], stride=stride)
    x = F.group_norm(x, groups, sd[name + ".1.weight"], sd[name + ".1.bias"], eps=1e-5)
    if up != 1:
        x = F.interpolate(x, scale_factor=up, mode="bilinear", align_corners=False)
    return F.relu(x) if relu else x


def irn_net_forward(x, sd, arch):
    """Net.forward (resnet50_irn.py:110-132 / vgg16_irn.py:192-212): x (N,3,S,S) -> (edge_out (N,1,h,w), dp_out (N,2,h,w))."""
    if arch == "m7":
        return m7_net_forward(x, sd)
    heads = RESNET50_HEADS if arch == "resnet50" else VGG16_HEADS
    xs = resnet50_stages(x, sd) if arch == "resnet50" else vgg16_stages(x, sd)
    e = [_head(xs[src - 1], sd, n, st, g, up) for n, src, st, _, g, up in heads["edge"]]
    h2, w2 = e[1].shape[2], e[1].shape[3]
    edge_out = F.conv2d(torch.cat([e[0], e[1]] + [t[..., :h2, :w2] for t in e[2:]], dim=1), sd["fc_edge6.weight"],
                        sd["fc_edge6.bias"])
    d = [_head(xs[src - 1], sd, n, st, g, up) for n, src, st, _, g, up in heads["dp"]]
    h3, w3 = d[2].shape[2], d[2].shape[3]
    dp_up3 = _head(torch.cat([d[2], d[3][..., :h3, :w3], d[4][..., :h3, :w3]], dim=1), sd, "fc_dp6", 1, 16, 2)
    dp_up3 = dp_up3[..., :d[1].shape[2], :d[1].shape[3]]
    hid = _head(torch.cat([d[0], d[1], dp_up3], dim=1), sd, "fc_dp7", 1, 16, 1)
    dp_out = F.conv2d(hid, sd["fc_dp7.3.weight"]) - sd["mean_shift.running_mean"].view(1, 2, 1, 1)  # MeanShift, eval
    return edge_out, dp_out


def edge_displacement_forward(x, sd, arch, crop_size=512, stride=4):
    """EdgeDisplacement.forward (resnet50_irn.py:218-232): x (2,3,h,w) -> (edge (1,fh,fw), dp (2,fh,fw))."""
    fh, fw = (x.size(2) - 1) // stride + 1, (x.size(3) - 1) // stride + 1
    x = F.pad(x, [0, crop_size - x.size(3), 0, crop_size - x.size(2)])
    edge_out, dp_out = irn_net_forward(x, sd, arch)
    edge_out = edge_out[..., :fh, :fw]
    dp_out = dp_out[..., :fh, :fw]
    edge_out = torch.sigmoid(edge_out[0] / 2 + edge_out[1].flip(-1) / 2)
    return edge_out, dp_out[0]
