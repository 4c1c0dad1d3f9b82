"""TEST INFRASTRUCTURE ONLY (oracle) -- torch-CPU fp32 restatement of the reference's CAM path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Each function cites the reference lines it follows (paths relative to /root/reference).
The ResNet50 restatement is pinned against the reference's own module
(03b_irn/net/resnet50.py, imported in this container by oracle/gen_golden.py) through
the committed fixtures under tests/golden/.

Everything here is a floating-point kernel restated with plain torch fp32 ops -- the
"plain PyTorch fp32 reference" the numerics tests compare the HIP kernels against.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

RESNET_BLOCKS = (3, 4, 6, 3)
RESNET_PLANES = (64, 128, 256, 512)
RESNET_CAM_STRIDES = (2, 2, 2, 1)  # 03b_irn/net/resnet50_cam.py:15

VGG16_CFG = [("layer1", [64, 64, "M"]), ("layer2", [128, 128, "M"]), ("layer3", [256, 256, 256, "M"]),
             ("layer4", [512, 512, 512, 512, 512, 512]), ("layer5", [1024, "D", 1024, "D"])]  # net/vgg16.py:44
M7_CFG = [("layer1", [64, 64, "M"]), ("layer2", [128, 128, "M"]), ("layer3_p1", [256, 256, 256])]  # net/m7.py:41


# ---------------------------------------------------------------------------------------------
# synthetic, seeded weights (there are no pretrained weights offline)
# ---------------------------------------------------------------------------------------------
def _bn_params(g, c, prefix, sd):
    sd[prefix + ".weight"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)
    sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_var"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)


def _conv_w(g, cout, cin, k, gain=1.0):
    fan_in = cin * k * k
    return torch.randn(cout, cin, k, k, generator=g) * (gain * math.sqrt(2.0 / fan_in))


def make_resnet50_cam_state_dict(num_classes=20, seed=0):
    """State dict with the keys of resnet50_cam.Net (`resnet50.*`, `classifier.weight`)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd["resnet50.conv1.weight"] = _conv_w(g, 64, 3, 7)
    _bn_params(g, 64, "resnet50.bn1", sd)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(zip(RESNET_PLANES, RESNET_BLOCKS, (1,) + RESNET_CAM_STRIDES[1:])):
        for bi in range(blocks):
            pre = "resnet50.layer%d.%d" % (li + 1, bi)
            s = stride if bi == 0 else 1
            sd[pre + ".conv1.weight"] = _conv_w(g, planes, inplanes, 1)
            _bn_params(g, planes, pre + ".bn1", sd)
            sd[pre + ".conv2.weight"] = _conv_w(g, planes, planes, 3)
            _bn_params(g, planes, pre + ".bn2", sd)
            # keep the residual branch small so activations stay O(1) through 16 blocks
            sd[pre + ".conv3.weight"] = _conv_w(g, planes * 4, planes, 1, gain=0.5)
            _bn_params(g, planes * 4, pre + ".bn3", sd)
            if bi == 0 and (s != 1 or inplanes != planes * 4):
                sd[pre + ".downsample.0.weight"] = _conv_w(g, planes * 4, inplanes, 1, gain=0.7)
                _bn_params(g, planes * 4, pre + ".downsample.1", sd)
            inplanes = planes * 4
    sd["classifier.weight"] = torch.randn(num_classes, 2048, 1, 1, generator=g) * 0.01
    return sd


def make_plain_state_dict(root, cfg, num_classes, batchnorm, seed=0, feat=None):
    """State dict of a common_cnn.make_layers stack (vgg16 / m7), keys `<root>.<layer>.<idx>.*`."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    cin = 3
    for lname, layer in cfg:
        idx = 0
        for v in layer:
            if v in ("M", "D"):
                idx += 1
                continue
            key = "%s.%s.%d" % (root, lname, idx)
            sd[key + ".weight"] = _conv_w(g, v, cin, 3)
            sd[key + ".bias"] = torch.randn(v, generator=g) * 0.05
            if batchnorm:
                _bn_params(g, v, "%s.%s.%d" % (root, lname, idx + 2), sd)
                idx += 3
            else:
                idx += 2
            cin = v
    sd[root + ".classifier.0.weight"] = torch.randn(num_classes, cin, generator=g) * 0.05
    sd[root + ".classifier.0.bias"] = torch.randn(num_classes, generator=g) * 0.05
    return sd


# ---------------------------------------------------------------------------------------------
# ResNet50 CAM
# ---------------------------------------------------------------------------------------------
def _fixed_bn(x, sd, pre, eps=1e-5):
    # FixedBatchNorm.forward, 03b_irn/net/resnet50.py:11-14
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"],
                        sd[pre + ".bias"], training=False, eps=eps)


def _bottleneck(x, sd, pre, stride):
    # Bottleneck.forward, 03b_irn/net/resnet50.py:34-54 (stride on conv2, :24)
    out = F.relu(_fixed_bn(F.conv2d(x, sd[pre + ".conv1.weight"]), sd, pre + ".bn1"))
    out = F.relu(_fixed_bn(F.conv2d(out, sd[pre + ".conv2.weight"], stride=stride, padding=1), sd, pre + ".bn2"))
    out = _fixed_bn(F.conv2d(out, sd[pre + ".conv3.weight"]), sd, pre + ".bn3")
    if pre + ".downsample.0.weight" in sd:
        residual = _fixed_bn(F.conv2d(x, sd[pre + ".downsample.0.weight"], stride=stride), sd,
                             pre + ".downsample.1")
    else:
        residual = x
    return F.relu(out + residual)


def resnet50_features(x, sd):
    """stage1..stage4 of resnet50_cam.Net (resnet50_cam.py:17-20; resnet50.py:62-64,96-104)."""
    x = F.conv2d(x, sd["resnet50.conv1.weight"], stride=RESNET_CAM_STRIDES[0], padding=3)
    x = F.relu(_fixed_bn(x, sd, "resnet50.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (blocks, stride) in enumerate(zip(RESNET_BLOCKS, (1,) + RESNET_CAM_STRIDES[1:])):
        for bi in range(blocks):
            x = _bottleneck(x, sd, "resnet50.layer%d.%d" % (li + 1, bi), stride if bi == 0 else 1)
    return x


def resnet50_cam_forward(x, sd):
    """CAM.forward, 03b_irn/net/resnet50_cam.py:55-70.  x: (2,3,S,S) -> (C,h,w)."""
    x = resnet50_features(x, sd)
    x = F.conv2d(x, sd["classifier.weight"])
    x = F.relu(x)
    x = x[0] + x[1].flip(-1)
    return x


# ---------------------------------------------------------------------------------------------
# VGG16 / M7 (common_cnn.make_layers)
# ---------------------------------------------------------------------------------------------
def last_bn_key(sd, root, cfg):
    """Key prefix of the BatchNorm that follows the LAST conv of a make_layers stack, or None (ADP VGG16: no BN)."""
    key = None
    for lname, layer in cfg:
        idx = 0
        for v in layer:
            if v in ("M", "D"):
                idx += 1
                continue
            bn = "%s.%s.%d" % (root, lname, idx + 2)
            key = bn if bn + ".running_mean" in sd else None
            idx += 3 if key else 2
    return key


def plain_features(x, sd, root, cfg, return_pre_bn=False):
    """make_layers stacks, 03b_irn/net/common_cnn.py:128-141: conv(bias) -> ReLU -> BatchNorm(eps=1e-3).
    return_pre_bn: also return the last conv's post-ReLU, PRE-BatchNorm activation -- the output of the Keras layer
    `find_final_layer` names (the layer after the last Conv2D = its Activation, 02_cues/utilities.py:42-58, "activation_7"),
    which is what the Keras-side drivers contract with alpha (02_cues/utilities.py:129-133, 03c_hsn/utilities.py:259-262)."""
    pre = None
    for lname, layer in cfg:
        idx = 0
        for v in layer:
            if v == "M":
                x = F.max_pool2d(x, kernel_size=2, stride=2)
                idx += 1
            elif v == "D":
                idx += 1  # nn.Dropout in eval()
            else:
                key = "%s.%s.%d" % (root, lname, idx)
                x = F.relu(F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], padding=1))
                pre = x
                bn = "%s.%s.%d" % (root, lname, idx + 2)
                if bn + ".running_mean" in sd:
                    x = _fixed_bn(x, sd, bn, eps=1e-3)
                    idx += 3
                else:
                    idx += 2
    return (x, pre) if return_pre_bn else x


def keras_conv_val(x, sd, root, cfg):
    """`conv_func([images])[0]` of the Keras drivers: the final Activation's output, NHWC (02_cues/utilities.py:129-132)."""
    with torch.no_grad():
        _, pre = plain_features(x, sd, root, cfg, return_pre_bn=True)
    return pre.permute(0, 2, 3, 1).contiguous()


def vgg16_cam_forward(x, sd, num_classes):
    """vgg16_cam.CAM.forward lines 26-50 (03b_irn/net/vgg16_cam.py): returns (cam (C,h,w), score (C,))."""
    x = plain_features(x, sd, "vgg16", VGG16_CFG)
    y = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    y = torch.sigmoid(F.linear(y, sd["vgg16.classifier.0.weight"], sd.get("vgg16.classifier.0.bias")))[0]
    cam = F.conv2d(x, sd["vgg16.classifier.0.weight"][:num_classes].unsqueeze(-1).unsqueeze(-1))
    cam = F.relu(cam)
    cam = cam[0] + cam[1].flip(-1)
    return cam, y[:num_classes]


def m7_cam_forward(x, sd, gradcam_weights, num_classes):
    """m7_cam.CAM.forward lines 26-47 (03b_irn/net/m7_cam.py)."""
    x = plain_features(x, sd, "m7", M7_CFG)
    y = F.max_pool2d(x, kernel_size=2, stride=2)  # layer3_p2 = ['M', 'D']
    y = torch.flatten(F.adaptive_max_pool2d(y, (1, 1)), 1)
    y = torch.sigmoid(F.linear(y, sd["m7.classifier.0.weight"], sd.get("m7.classifier.0.bias")))[0]
    w = gradcam_weights.float()  # (F, C)
    cam = F.conv2d(x, w.transpose(1, 0).unsqueeze(-1).unsqueeze(-1))
    cam = F.relu(cam)
    cam = cam[0] + cam[1].flip(-1)
    return cam, y[:num_classes]


def grad_cam_weights(sd, root, cfg, S, num_classes):
    """get_grad_cam_weights (02_cues/utilities.py:60-99; 03b_irn/net/common_cnn.py:84-121) on the
    restated torch net: alpha[:, c] = mean_hw normalize(d logit_c / d A) on a zeros image, where A is the output of the
    layer AFTER the last Conv2D (find_final_layer) -- its ReLU Activation, i.e. the PRE-BatchNorm tensor of the
    conv -> ReLU -> BatchNorm order (common_cnn.py:138): the gradient passes through the inference-mode BatchNorm."""
    x = torch.zeros(1, 3, S, S)
    with torch.no_grad():
        _, pre = plain_features(x, sd, root, cfg, return_pre_bn=True)
    A = pre.detach().requires_grad_(True)
    bn = last_bn_key(sd, root, cfg)
    feat = _fixed_bn(A, sd, bn, eps=1e-3) if bn else A
    if root == "m7":
        pooled = torch.flatten(F.adaptive_max_pool2d(F.max_pool2d(feat, 2, 2), (1, 1)), 1)
    else:
        pooled = torch.flatten(F.adaptive_avg_pool2d(feat, (1, 1)), 1)
    logits = F.linear(pooled, sd[root + ".classifier.0.weight"], sd.get(root + ".classifier.0.bias"))[0]
    alpha = np.zeros((A.shape[1], num_classes))
    for c in range(num_classes):
        (g,) = torch.autograd.grad(logits[c], A, retain_graph=True)
        g = g / (torch.sqrt(torch.mean(g * g)) + 1e-5)
        alpha[:, c] = g[0].mean(dim=(1, 2)).numpy()
    return alpha


# ---------------------------------------------------------------------------------------------
# make_cam tail
# ---------------------------------------------------------------------------------------------
def get_strided_size(orig_size, stride):
    # misc.imutils.get_strided_size (not in tree; upstream jiwoon-ahn/irn), SURVEY.md Appendix A
    return ((orig_size[0] - 1) // stride + 1, (orig_size[1] - 1) // stride + 1)


def get_strided_up_size(orig_size, stride):
    s = get_strided_size(orig_size, stride)
    return (s[0] * stride, s[1] * stride)


def make_cam_tail(cam, size, valid_cat):
    """make_cam._work, 03b_irn/step/make_cam.py:41-42,62-76.

    cam: (C,h,w) tensor, or a list of them (one per args.cam_scales entry: every scale is interpolated, THEN summed, the
    reference's order); size: (H0,W0); valid_cat: LongTensor of class indices.
    Returns (strided_cam (K,h4,w4), highres_cam (K,H0,W0)) float32 tensors.
    """
    strided_size = get_strided_size(size, 4)
    strided_up_size = get_strided_up_size(size, 16)
    outputs = list(cam) if isinstance(cam, (list, tuple)) else [cam]
    strided_cam = torch.sum(torch.stack(
        [F.interpolate(torch.unsqueeze(o, 0), strided_size, mode="bilinear", align_corners=False)[0]
         for o in outputs]), 0)
    highres_cam = [F.interpolate(torch.unsqueeze(o, 1), strided_up_size, mode="bilinear", align_corners=False)
                   for o in outputs]
    highres_cam = torch.sum(torch.stack(tuple(highres_cam), 0), 0)[:, 0, :size[0], :size[1]]
    strided_cam = strided_cam[valid_cat]
    strided_cam = strided_cam / (F.adaptive_max_pool2d(strided_cam, (1, 1)) + 1e-5)
    highres_cam = highres_cam[valid_cat]
    highres_cam = highres_cam / (F.adaptive_max_pool2d(highres_cam, (1, 1)) + 1e-5)
    return strided_cam, highres_cam


def make_cam_image(x_pair, sd, size, label):
    """One image through the reference op sequence of make_cam._work (make_cam.py:36-82):
    x_pair (2,3,S,S) -> dict {"keys","cam","high_res"} exactly as np.save'd."""
    with torch.no_grad():
        cam = resnet50_cam_forward(x_pair, sd)
        valid_cat = torch.nonzero(label)[:, 0]
        if len(valid_cat) == 0:
            return {"keys": np.empty(0), "cam": np.empty(0), "high_res": np.empty(0)}
        s, h = make_cam_tail(cam, size, valid_cat)
    return {"keys": valid_cat.numpy(), "cam": s.numpy(), "high_res": h.numpy()}


# ---------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------
VOC_SIZES = [(375, 500)] * 6 + [(500, 375)] * 2 + [(333, 500)] + [(500, 500)]


def synth_image(rng, H, W):
    """uint8 RGB: soft-edged ellipses over a low-frequency gradient + noise."""
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


def resize_bilinear_f64(img, out_hw):
    """Stand-in for TorchvisionResize (03b_irn/voc12/dataloader.py:68-78: float64 cv2.resize, bilinear,
    half-pixel centres).  cv2 is absent offline, so parity of this resize itself is unpinned; the
    oracle and the build are always fed the same already-resized tensors."""
    H, W = img.shape[:2]
    oh, ow = out_hw
    if (H, W) == (oh, ow):
        return img.astype(np.float64)
    ys = np.clip((np.arange(oh) + 0.5) * H / oh - 0.5, 0, H - 1)
    xs = np.clip((np.arange(ow) + 0.5) * W / ow - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0)[:, None, None]; wx = (xs - x0)[None, :, None]
    im = img.astype(np.float64)
    top = im[y0][:, x0] * (1 - wx) + im[y0][:, x1] * wx
    bot = im[y1][:, x0] * (1 - wx) + im[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def normalize_int(img):
    """TorchvisionNormalize('int'), 03b_irn/voc12/dataloader.py:80-106: (x - (104,117,123)) / 255 on R,G,B."""
    out = np.empty(img.shape, np.float32)
    im = np.float32(img)
    for c, m in enumerate((104.0, 117.0, 123.0)):
        out[..., c] = (im[..., c] - m) / 255.0
    return out


def normalize_float(img):
    """TorchvisionNormalize('float'), 03b_irn/voc12/dataloader.py:80-106: (x / 255 - mean) / std with the ImageNet constants."""
    out = np.empty(img.shape, np.float32)
    im = np.float32(img)
    for c, (m, s) in enumerate(((0.485, 0.229), (0.456, 0.224), (0.406, 0.225))):
        out[..., c] = (im[..., c] / 255. - m) / s
    return out


def msf_pack(img_u8, outsize=(321, 321), norm_mode="int"):
    """VOC12ClassificationDatasetMSF.__getitem__ for scales=(1.0,), voc12/dataloader.py:225-246:
    resize -> normalise -> HWC_to_CHW -> stack([x, flip(x, -1)])  => float32 (2,3,S,S)."""
    # outsize None: TorchvisionResize is a no-op, the image keeps its own size (voc12/dataloader.py:74)
    normalize = {"int": normalize_int, "float": normalize_float}[norm_mode]
    x = normalize(resize_bilinear_f64(img_u8, outsize) if outsize is not None else np.asarray(img_u8, np.float64))
    x = np.transpose(x, (2, 0, 1))
    return np.stack([x, np.flip(x, -1)], axis=0).astype(np.float32)


ADP_INDS_X17 = [2, 3, 4, 6, 7, 8, 9, 12, 13, 14, 16, 17, 18, 21, 22, 23, 25, 26, 28, 29, 30, 32, 33, 35, 37, 38, 40, 45, 48,
                49, 50]  # 03b_irn/net/common_cam.py:26-29


def adp_modify(cam, img_orig, dataset, use_cls, x17=False):
    """CommonCAM._adp_modify_morph / _adp_modify_func, 03b_irn/net/common_cam.py:31-92 (called from vgg16_cam.py:49-58 /
    m7_cam.py:48-55 after the X1.7 class filter), in the reference's own numpy / scipy calls: cam float32 (C, h, w),
    img_orig uint8 (2, H0, W0, 3) -> the stack [background | (other) | cam[use_cls]].  cv2.resize (absent offline) is the
    bilinear resize pinned in tests/test_dataloaders_host.py."""
    import scipy.ndimage
    import scipy.special

    cam = np.asarray(cam, np.float32)
    if x17:
        cam = cam[ADP_INDS_X17]
    mean_img = np.asarray(img_orig[0], np.float32).mean(axis=2)                       # torch.mean(img_orig[0].float(), dim=2)
    bg = 0.75 * scipy.special.expit(4 * (mean_img - 240))                             # :36-42
    bg = scipy.ndimage.gaussian_filter(bg, sigma=2)                                    # :43
    if bg.shape != cam.shape[1:]:
        bg = resize_bilinear_f64(bg[..., None], cam.shape[1:])[..., 0]                 # :44 cv2.resize
    bg = bg.astype(np.float32)
    adipose = cam[[18, 19, 20]].max(axis=0)
    if dataset == "adp_morph":
        background = np.maximum(bg - adipose, 0)                                       # :47-51 relu
        return np.concatenate((background[None], cam[use_cls]), axis=0)                # :54
    background = bg - cam[[28, 29, 30]].max(axis=0)                                    # :74-78 (no relu)
    modified = np.concatenate((background[None], cam[use_cls]), axis=0)                # :81
    other = np.maximum(np.float32(0.05) * (1 - modified.max(axis=0)), adipose)         # :84-88
    return np.concatenate((modified[:1], other[None], modified[1:]), axis=0)          # :91
