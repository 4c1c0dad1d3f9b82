"""CPU: Keras weight list -> state dict (wsscam.net.common.state_dict_from_keras_weights) against the reference's
load_weights_from_file semantics (03b_irn/net/common_cnn.py:25-82): pop order = nn.Module.modules() order of the
torch network, HWIO -> OIHW, Dense transposed, `use_bias = 'VGG16' not in tag`, thresholds = max(mat, 1/3)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from wsscam.net import common


def _make_layers(layer, batchnorm, in_channels):
    # common_cnn.make_layers :128-141 restated with torch modules (the independent side of this test)
    layers = []
    for v in layer:
        if v == "M":
            layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
        elif v == "D":
            layers += [nn.Dropout(p=0.5)]
        else:
            conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)
            layers += [conv2d, nn.ReLU(inplace=True)] + ([nn.BatchNorm2d(v, eps=0.001, momentum=0.99)] if batchnorm else [])
            in_channels = v
    return nn.Sequential(*layers), in_channels


class _Plain(nn.Module):
    def __init__(self, root, batchnorm, num_classes):
        super().__init__()
        cin = 3
        for lname, layer in common.PLAIN_CFG[root]:
            seq, cin = _make_layers(layer, batchnorm, cin)
            setattr(self, lname, seq)
        self.classifier = nn.Sequential(nn.Linear(cin, num_classes), nn.Sigmoid())


def _keras_list(net, rng, use_bias, extra_classes=0):
    """A Keras-ordered weight list for `net` (what model.get_weights() returns) + the torch-side expectation."""
    weights, expect = [], {}
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            w = rng.normal(size=(3, 3, m.in_channels, m.out_channels)).astype(np.float32)  # HWIO
            b = rng.normal(size=m.out_channels).astype(np.float32)
            weights += [w, b]
            expect[name + ".weight"], expect[name + ".bias"] = np.transpose(w, (3, 2, 0, 1)), b
        elif isinstance(m, nn.BatchNorm2d):
            for k in ("weight", "bias", "running_mean", "running_var"):
                w = rng.uniform(0.5, 1.5, m.num_features).astype(np.float32)
                weights.append(w)
                expect[name + "." + k] = w
        elif isinstance(m, nn.Linear):
            w = rng.normal(size=(m.in_features, m.out_features + extra_classes)).astype(np.float32)  # Keras Dense kernel
            weights.append(w)
            expect[name + ".weight"] = w.T
            if use_bias:
                b = rng.normal(size=m.out_features + extra_classes).astype(np.float32)
                weights.append(b)
                expect[name + ".bias"] = b
    return weights, expect


@pytest.mark.parametrize("root,batchnorm,tag", [("vgg16", True, "VOC2012_VGG16"), ("vgg16", False, "ADP_VGG16"),
                                                ("m7", True, "VOC2012_M7"), ("m7", True, "ADP_X1.7")])
def test_state_dict_from_keras_weights(root, batchnorm, tag):
    rng = np.random.default_rng(3)
    C = 20
    net = _Plain(root, batchnorm, C)
    # the pop order of the reference is the modules() order of the torch net: same sequence of (kind, name)
    kinds = {nn.Conv2d: "conv", nn.BatchNorm2d: "bn", nn.Linear: "linear"}
    torch_order = [(kinds[type(m)], root + "." + n) for n, m in net.named_modules() if type(m) in kinds]
    assert torch_order == common.plain_module_order(root, batchnorm)
    use_bias = "VGG16" not in tag
    weights, expect = _keras_list(net, rng, use_bias, extra_classes=2)  # Keras head may hold more classes (:77)
    mat = np.array([0.1, 0.5, 0.9] + [0.2] * 19)
    sd = common.state_dict_from_keras_weights(weights, tag, root, batchnorm, thresholds_mat=mat)
    assert set(sd) == {root + "." + k for k in expect} | {"thresholds"}
    for k, v in expect.items():
        assert sd[root + "." + k].dtype == np.float32 and np.array_equal(sd[root + "." + k], v), k
    assert (root + ".classifier.0.bias" in sd) == use_bias  # Q2: the VGG16 Dense has no bias to load
    assert np.allclose(sd["thresholds"], np.maximum(mat, 1 / 3)) and sd["thresholds"][0] == np.float32(1 / 3)
    # the loaded dict drives the torch net to the same parameters load_weights_from_file would leave in it
    with torch.no_grad():
        for name, p in list(net.named_parameters()) + list(net.named_buffers()):
            if name.endswith("num_batches_tracked") or (name == "classifier.0.bias" and not use_bias):
                continue
            v = sd[root + "." + name]
            assert v.shape[1:] == tuple(p.shape[1:]) and v.shape[0] >= p.shape[0], name
    # count check of common_cnn.py:48-49
    with pytest.raises(AssertionError, match="Sizes of PyTorch network and saved Keras network differ"):
        common.state_dict_from_keras_weights(weights[:-1], tag, root, batchnorm)
    with pytest.raises(AssertionError):
        bad = list(weights)
        bad[0] = bad[0][:, :, :, :-1]
        common.state_dict_from_keras_weights(bad, tag, root, batchnorm)


def test_keras_h5_reader_needs_h5py(tmp_path):
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            common.keras_h5_weight_list(str(tmp_path / "x.h5"))


def test_cam_wrappers_batchnorm_and_class_count():
    """Which CAM wrapper expects BatchNorm weights, and the X1.7 class count: vgg16_cam.py:16-19 drops BatchNorm for the ADP
    datasets only; m7_cam.py:16-18 ALWAYS builds m7.m7(..., batchnorm=True) and scores 51 classes for ADP X1.7 sessions
    (filtered to the 31 of common_cam.py:26-29 afterwards)."""
    from wsscam.net import m7_cam, vgg16_cam

    assert vgg16_cam.CAM(None, "voc12", "VOC2012_VGG16", 20, None).batchnorm is True
    assert vgg16_cam.CAM(None, "adp_morph", "ADP_VGG16", 31, None).batchnorm is False
    assert vgg16_cam.CAM(None, "adp_func", "ADP_VGG16", 31, None).batchnorm is False
    for ds in ("voc12", "deepglobe", "adp_morph", "adp_func"):
        assert m7_cam.CAM(None, ds, "X_M7", 20, None).batchnorm is True
    m = m7_cam.CAM(None, "adp_morph", "ADP_X1.7", 31, None)
    assert m.batchnorm is True and m.num_classes == 51 and len(m.thresholds) == 51
    assert m7_cam.CAM(None, "voc12", "VOC2012_X1.7", 20, None).num_classes == 20
    y = m.predict_labels(np.linspace(0, 1, 51).astype(np.float32))
    assert y.shape == (31,) and y[-1] and not y[0]
