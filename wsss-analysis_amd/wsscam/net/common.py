"""Shared host logic of the CAM network wrappers (weights -> wsc_net, batched forward)."""
import numpy as np

from .. import _lib


def _to_numpy_sd(state_dict):
    out = {}
    for k, v in state_dict.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype.kind not in "fiu" or v.ndim > 4:
            continue
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class DeviceCAMBase:
    """Common plumbing: a lazily created wsc_ctx + wsc_net, numpy/torch in, same kind out."""

    arch = None
    # the fp32-class mode: the reference's arithmetic is fp32 and the package is a drop-in (PREC_F16 / PREC_BF16: fast 16-bit
    # modes, 1.5e-2 / 8e-2 on the normalised CAM maps -- args.cam_precision or the constructor's `precision`)
    precision = _lib.PREC_F16X3

    def __init__(self, num_classes, precision=None):
        self.num_classes = num_classes
        if precision is not None:
            self.precision = precision
        self._sd = None
        self._ctx = None
        self._net = None
        self._device = 0
        self.training = False

    # -- nn.Module-like surface used by make_cam.run (make_cam.py:98-100, 33) ---------------
    def load_state_dict(self, state_dict, strict=True):
        self._sd = _to_numpy_sd(state_dict)
        self._release_net()
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self, device=None):
        if device is not None:
            self._device = int(device)
        self._ensure_net()
        return self

    def _release_net(self):
        if self._net is not None:
            self._net.close()
            self._net = None

    def _ensure_net(self):
        if self._net is None:
            if self._sd is None:
                raise RuntimeError("load_state_dict() must be called before the network is used")
            if self._ctx is None:
                self._ctx = _lib.Context(self._device)
            self._net = _lib.Net(self._ctx, self.arch, self._extra_tensors(self._sd), self.num_classes,
                                 self.precision)
        return self._net

    def _extra_tensors(self, sd):
        return sd

    @property
    def ctx(self):
        self._ensure_net()
        return self._ctx

    def cam_size(self, S):
        return self._ensure_net().cam_size(S)

    def cam_size_hw(self, H, W):
        return self._ensure_net().cam_size_hw(H, W)

    def gradcam_net(self, weights, ctx=None, pre_bn=True):
        """(wsc_net with `weights` (F x C Grad-CAM alpha) as its 1x1 head, ctx); cached per alpha.  `ctx`: run on
        this context instead of the model's own (two models of one driver share a stream and its buffers).
        pre_bn=True (the Keras drivers, 02_cues/utilities.py:129-133 and 03c_hsn/utilities.py:259-262): alpha is
        contracted with the final Activation's output, i.e. the tensor BEFORE the last BatchNorm -- realised as an
        equivalent head on the post-BN feature map (`pre_bn_head`).  pre_bn=False contracts with the post-BN map as
        the torch port's m7_cam.py:45-46 does."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        key = (w.shape, hash(w.tobytes()), bool(pre_bn))
        cache = self.__dict__.setdefault("_gradcam_nets", {})
        if key not in cache:
            if self._ctx is None:
                self._ctx = _lib.Context(self._device)
            sd = dict(self._sd)
            affine = last_bn_affine(sd, self.root) if pre_bn and getattr(self, "root", None) in PLAIN_CFG else None
            hw, hb = pre_bn_head(w, affine)
            sd["gradcam_weights"] = np.ascontiguousarray(hw, dtype=np.float32)
            sd["gradcam_bias"] = np.ascontiguousarray(hb, dtype=np.float32)
            cache[key] = _lib.Net(self._ctx, self.arch, sd, w.shape[1], self.precision)
        return cache[key], (ctx or self._ctx)

    # -- batched device forward ---------------------------------------------------------------
    def forward_batch_device(self, x_dev, B, S, cam_dev, score_dev=None, SW=None):
        """x_dev float32 [B][2][3][S][SW or S] -> cam_dev float32 [B][C][h][w] (device pointers/buffers)."""
        if SW is None or SW == S:
            self._ensure_net().forward_cam(x_dev, B, S, cam_dev, score_dev)
        else:
            self._ensure_net().forward_cam_hw(x_dev, B, S, SW, cam_dev, score_dev)

    def forward_batch(self, x, want_score=False):
        """x: numpy/torch float32 (B,2,3,S,S) on host -> numpy cam (B,C,h,w) [, score (B,C)]."""
        net = self._ensure_net()
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        xn = np.ascontiguousarray(xn, dtype=np.float32)
        assert xn.ndim == 5 and xn.shape[1] == 2 and xn.shape[2] == 3 and xn.shape[3] == xn.shape[4], xn.shape
        B, S = xn.shape[0], xn.shape[3]
        h = net.cam_size(S)
        ctx = self._ctx
        x_dev = ctx.to_device(xn)
        cam_dev = ctx.alloc(B * self.num_classes * h * h * 4)
        score_dev = ctx.alloc(B * self.num_classes * 4) if want_score else None
        net.forward_cam(x_dev, B, S, cam_dev, score_dev)
        cam = ctx.to_host(cam_dev, (B, self.num_classes, h, h), np.float32)
        score = ctx.to_host(score_dev, (B, self.num_classes), np.float32) if want_score else None
        return (cam, score) if want_score else cam


# ---- Keras weight list -> state dict (03b_irn/net/common_cnn.py:25-82) ----------------------------------------
# layer tables of the two plain stacks (net/vgg16.py:44, net/m7.py:41): 'M' MaxPool2d(2,2), 'D' Dropout
PLAIN_CFG = {
    "vgg16": [("layer1", [64, 64, "M"]), ("layer2", [128, 128, "M"]), ("layer3", [256, 256, 256, "M"]),
              ("layer4", [512, 512, 512, 512, 512, 512]), ("layer5", [1024, "D", 1024, "D"])],
    "m7": [("layer1", [64, 64, "M"]), ("layer2", [128, 128, "M"]), ("layer3_p1", [256, 256, 256])],
}


def plain_module_order(root, batchnorm):
    """(kind, key prefix) of every parameterised module of VGG / MNet in `nn.Module.modules()` order, which is the
    order load_weights_from_file pops the Keras list in (common_cnn.py:53-82): make_layers emits
    conv -> ReLU -> BatchNorm per entry (common_cnn.py:137-141), the classifier's Linear comes last."""
    order = []
    for lname, layer in PLAIN_CFG[root]:
        idx = 0
        for v in layer:
            if v in ("M", "D"):
                idx += 1
                continue
            order.append(("conv", "%s.%s.%d" % (root, lname, idx)))
            if batchnorm:
                order.append(("bn", "%s.%s.%d" % (root, lname, idx + 2)))
                idx += 3
            else:
                idx += 2
    order.append(("linear", root + ".classifier.0"))
    return order


def state_dict_from_keras_weights(weights, tag, root="vgg16", batchnorm=True, thresholds_mat=None):
    """`model.get_weights()` of the pretrained Keras CNN -> the state dict the device nets take.

    Restates CommonCNN.load_weights_from_file / load_thresholds_from_file (common_cnn.py:25-82, 123-125):
      * Conv2D kernels HWIO -> OIHW (`HWCD_to_DCHW`, :42-43), followed by their bias;
      * BatchNormalization as [gamma, beta, moving_mean, moving_variance] -> weight, bias, running_mean, running_var;
      * Dense kernel transposed to (classes, features); its bias is loaded only when 'VGG16' is NOT in the tag
        (`use_bias`, :44 -- the Keras Dense of the VGG16 models has none).  The reference then keeps nn.Linear's
        random-initialised bias; here the key is simply absent and the device net uses a zero bias;
      * the count check of :48-49 and every shape check are kept (AssertionError, as in the reference);
      * thresholds = max(optimalScoreThresh, 1/3) (:39), stored under "thresholds".
    `weights` is consumed front to back like the reference's `weights.pop(0)`; it is not modified.
    """
    weights = [np.asarray(w) for w in weights]
    use_bias = "VGG16" not in tag
    order = plain_module_order(root, batchnorm)
    n_conv = sum(1 for k, _ in order if k == "conv")
    n_bn = sum(1 for k, _ in order if k == "bn")
    n_lin = sum(1 for k, _ in order if k == "linear")
    assert 2 * n_conv + 4 * n_bn + (1 + use_bias) * n_lin == len(weights), \
        "Sizes of PyTorch network and saved Keras network differ!"
    sd = {}
    pos = 0
    cin = 3
    for kind, key in order:
        if kind == "conv":
            w = np.transpose(weights[pos], (3, 2, 0, 1))
            b = weights[pos + 1]
            pos += 2
            assert w.ndim == 4 and w.shape[1] == cin and w.shape[2:] == (3, 3), (key, w.shape)
            assert b.shape == (w.shape[0],), (key, b.shape)
            sd[key + ".weight"], sd[key + ".bias"] = w, b
            cin = w.shape[0]
        elif kind == "bn":
            for name in ("weight", "bias", "running_mean", "running_var"):
                w = weights[pos]
                pos += 1
                assert w.shape == (cin,), (key, name, w.shape)
                sd[key + "." + name] = w
        else:
            w = np.transpose(weights[pos])
            pos += 1
            assert w.ndim == 2 and w.shape[1] == cin, (key, w.shape)
            sd[key + ".weight"] = w
            if use_bias:
                b = weights[pos]
                pos += 1
                assert b.shape == (w.shape[0],), (key, b.shape)
                sd[key + ".bias"] = b
    if thresholds_mat is not None:
        sd["thresholds"] = np.maximum(np.asarray(thresholds_mat, dtype=np.float64).reshape(-1), 1 / 3)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}


def keras_h5_weight_list(weights_path):
    """The list `model.get_weights()` returns, read straight from a Keras 2 `.h5` weight file (layer_names /
    weight_names attributes give the order): needs h5py, which is only required on this one path."""
    try:
        import h5py
    except ImportError as e:  # no silent fallback: the caller asked for pretrained weights
        raise RuntimeError("reading %s needs h5py (pass state_dict / a Keras weight list instead)" % weights_path) from e
    out = []
    with h5py.File(weights_path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        for lname in g.attrs["layer_names"]:
            lname = lname.decode() if isinstance(lname, bytes) else lname
            for wname in g[lname].attrs["weight_names"]:
                wname = wname.decode() if isinstance(wname, bytes) else wname
                out.append(np.asarray(g[lname][wname]))
    return out


def load_pretrained(model_dir, tag, root="vgg16", batchnorm=True):
    """CommonCNN._load_pretrained (common_cnn.py:25-41): <model_dir>/<tag>/<tag>.h5 + <tag>.mat -> state dict."""
    import os

    import scipy.io

    weights = keras_h5_weight_list(os.path.join(model_dir, tag, tag + ".h5"))
    mat = scipy.io.loadmat(os.path.join(model_dir, tag, tag + ".mat")).get("optimalScoreThresh")[0]
    return state_dict_from_keras_weights(weights, tag, root, batchnorm, mat)


def last_bn_affine(sd, root):
    """(scale, shift) of the BatchNorm that follows the LAST conv of a make_layers stack as the per-channel map
    post = scale * pre + shift (inference mode, eps = 1e-3: common_cnn.py:138), or None when the stack has no
    BatchNorm there (the ADP VGG16 models, vgg16_cam.py:16-17).  `pre` is the tensor the Keras drivers call the final
    layer's output: find_final_layer returns the layer AFTER the last Conv2D, its ReLU Activation ("activation_7",
    02_cues/utilities.py:42-58) -- and the reference's layer order is Conv2D -> Activation -> BatchNormalization
    (make_layers' "# reversed", the get_weights() pop order of load_weights_from_file)."""
    key = None
    for lname, layer in PLAIN_CFG[root]:
        idx = 0
        for v in layer:
            if v in ("M", "D"):
                idx += 1
                continue
            bn = "%s.%s.%d" % (root, lname, idx + 2)
            key = bn if bn + ".running_mean" in sd else None
            idx += 3 if key else 2
    if key is None:
        return None
    eps = float(np.asarray(sd[key + ".eps"]).reshape(-1)[0]) if key + ".eps" in sd else 1e-3
    scale = np.asarray(sd[key + ".weight"], np.float64) / np.sqrt(np.asarray(sd[key + ".running_var"], np.float64) + eps)
    shift = np.asarray(sd[key + ".bias"], np.float64) - np.asarray(sd[key + ".running_mean"], np.float64) * scale
    return scale, shift


def grad_cam_alpha(W, h, w, pool, should_normalize=True, bn_scale=None):
    """Grad-CAM weights alpha (F, C) of a [BatchNorm ->] `pool` -> Linear classifier head on an h x w x F activation A
    (02_cues/utilities.py:60-99, common_cnn.py:84-121): g = d y_c / d A, g <- g / (sqrt(mean(g^2)) + 1e-5),
    alpha[:, c] = mean_{h,w} g.  A is the final conv's ReLU output; when the stack carries a BatchNorm between A and the
    pooling (`bn_scale` = gamma / sqrt(running_var + eps), see last_bn_affine) the gradient passes through it:
    g = W[c, f] * bn_scale[f] * (d pool / d feature).  pool = "avg": 1 / (h w) everywhere; "max": 1 at the one position
    the pooling selected in channel f, else 0 (which position wins does not matter to the mean or to the RMS)."""
    W = np.asarray(W, dtype=np.float64)  # (C, F)
    if bn_scale is not None:
        W = W * np.asarray(bn_scale, dtype=np.float64)[None, :]
    F = W.shape[1]
    hw = float(h * w)
    if pool == "avg":
        mean_g = W / hw                                                     # spatial mean of a constant map
        rms = np.sqrt(np.sum(np.float32(W / hw) ** 2, axis=1, keepdims=True, dtype=np.float64) / F)
    elif pool == "max":
        mean_g = W / hw                                                     # one entry W[c, f] averaged over h w positions
        rms = np.sqrt(np.sum(np.float32(W) ** 2, axis=1, keepdims=True, dtype=np.float64) / (F * hw))
    else:
        raise ValueError("pool must be 'avg' or 'max'")
    g = mean_g / (rms + 1e-5) if should_normalize else mean_g
    return np.ascontiguousarray(g.T)


def pre_bn_head(alpha, affine):
    """Head (weights (F, C), bias (C,)) that contracts alpha with the PRE-BatchNorm activation while the device stack
    hands the head its post-BatchNorm feature map: post = s * pre + t  =>  sum_f pre_f alpha_fc =
    sum_f post_f (alpha_fc / s_f) - sum_f t_f alpha_fc / s_f.  Exact in real arithmetic; a zero BatchNorm scale makes
    the pre-BN activation unrecoverable and is an error."""
    alpha = np.asarray(alpha, dtype=np.float64)
    if affine is None:
        return alpha, np.zeros(alpha.shape[1])
    s, t = affine
    s = np.asarray(s, dtype=np.float64)
    if np.any(s == 0):
        raise ValueError("BatchNorm scale of the final layer has zeros: the pre-BN activation cannot be recovered")
    # a channel whose scale is tiny next to the others (|s_f| < 1e-6 max|s|) would get head weights of 1e6 alpha and beyond
    # the half range, and its t / s cancellation amplifies the 16-bit feature quantisation by the same factor; alpha from
    # grad_cam_alpha carries the factor s_f itself, so such a channel contributes ~ s_f^2: dropped (weight and bias term 0)
    small = np.abs(s) < 1e-6 * np.abs(s).max()
    w = np.where(small[:, None], 0.0, alpha / np.where(small, 1.0, s)[:, None])
    return w, -(t[:, None] * w).sum(axis=0)
