"""Grad-CAM cue generation on the device -- mirror of 02_cues/utilities.py.

`input_model` is a wsscam CAM wrapper (wsscam.net.vgg16_cam.CAM / m7_cam.CAM holding a torch-style
state dict) where the reference passes a Keras Sequential model; every other argument keeps its
meaning.  The conv stack and the `einsum('ijkl,lm->ijkm')` contraction run in ONE pass of libwsscam
(wsc_net_forward_gradcam); the reference evaluates the CNN twice per batch."""
import numpy as np

from .. import _lib


def find_final_layer(model):
    """02_cues/utilities.py:42-58: name of the layer after the last Conv2D.  The device networks expose
    exactly one such tap (the last conv feature map), identified by this constant."""
    return "final_conv_activation"


def get_grad_cam_weights(input_model, final_layer, dummy_image, should_normalize=True):
    """02_cues/utilities.py:60-99 (twin: 03b_irn/net/common_cnn.py:84-121): alpha[:, c] = mean_{h,w} normalize(d y_c / d A)
    on `dummy_image`, y_c the pre-sigmoid logit, A = the output of `final_layer` -- the layer after the last Conv2D, its ReLU
    Activation, which sits BEFORE the last BatchNorm in the reference's Conv2D -> Activation -> BatchNormalization order
    (common_cnn.py:138 "# reversed"), so K.gradients passes through the inference-mode BatchNorm:
    d y_c / d A = W[c, f] * gamma_f / sqrt(var_f + 1e-3) * (d pool / d feature).

    Both classifier heads of the reference's networks give a closed form (net.common.grad_cam_alpha):
      * GAP + Linear (modified VGG16 / X1.7, net/vgg16.py:17-22): d pool = 1 / (h w) at every position;
      * MaxPool + global max + Linear (M7, net/m7.py:15-21): d pool = 1 at the ONE position the pooling selected in
        channel f and zero elsewhere -- which position wins the ties of the all-zeros image does not matter, because
        alpha is the spatial MEAN of the normalised gradient.  (A framework that splits the gradient of a tied maximum
        evenly over the ties, as TF's reduce_max does, keeps the mean but changes the RMS normaliser; the Keras
        architecture files are not in the tree, so this is the torch port's reading, m7.py:17 AdaptiveMaxPool2d.)
    Equal to torch.autograd on the restated nets up to round-off (tests/test_cues_host.py)."""
    from ..net.common import PLAIN_CFG, grad_cam_alpha, last_bn_affine

    sd = input_model._sd
    root = input_model.root
    W = np.asarray(sd[root + ".classifier.0.weight"], dtype=np.float64)  # (C, F)
    S = int(dummy_image.shape[1])
    h = S
    for _, layer in PLAIN_CFG[root]:
        h //= 2 ** sum(1 for v in layer if v == "M")
    affine = last_bn_affine(sd, root)
    return grad_cam_alpha(W, h, h, "max" if root == "m7" else "avg", should_normalize,
                          bn_scale=None if affine is None else affine[0])


def _to_nchw(images):
    x = np.asarray(images, dtype=np.float32)
    assert x.ndim == 4 and x.shape[-1] == 3, "images must be (B, H, W, 3)"
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def conv_and_cams(input_model, weights, images, relu, want_scores=False):
    """One device pass: (cams (B,h,w,C) float32 = [relu] einsum(conv_val, weights), scores (B,C) or None); conv_val is the
    final Activation's output (pre-BatchNorm, `gradcam_net(pre_bn=True)`), as K.function([input], [conv_output]) gives it."""
    net, ctx = input_model.gradcam_net(weights)
    x = _to_nchw(images)
    B, S = x.shape[0], x.shape[2]
    h = net.cam_size(S)
    C = weights.shape[1]
    x_dev = ctx.to_device(x)
    cams_dev = ctx.alloc(B * h * h * C * 4)
    score_dev = ctx.alloc(B * C * 4) if want_scores else None
    net.forward_gradcam(x_dev, B, S, relu, cams_dev, score_dev)
    cams = ctx.to_host(cams_dev, (B, h, h, C), np.float32)
    scores = ctx.to_host(score_dev, (B, C), np.float32) if want_scores else None
    return cams, scores


def _upsample_nhwc(ctx, cams, size):
    """cv2.resize(cams[i, :, :, j], size) for every (i, j): bilinear, half-pixel centres."""
    B, h, w, C = cams.shape
    src = np.ascontiguousarray(np.transpose(cams, (0, 3, 1, 2)).reshape(B * C, h, w), dtype=np.float32)
    dst = ctx.alloc(B * C * size[0] * size[1] * 4)
    _lib.bilinear_resize(ctx, ctx.to_device(src), B * C, h, w, dst, size[0], size[1])
    out = ctx.to_host(dst, (B, C, size[0], size[1]), np.float32)
    return np.transpose(out, (0, 2, 3, 1))


def grad_cam(input_model, weights, images, is_pass_threshold, final_layer, keep_inds, orig_sz=[224, 224],
             should_upsample=False):
    """02_cues/utilities.py:101-144 -> (B, h, w, C_keep) thresholded Grad-CAMs (float64 like the reference)."""
    cams, _ = conv_and_cams(input_model, np.asarray(weights), images, relu=True)
    cams = cams[:, :, :, keep_inds]
    if should_upsample:
        cams = _upsample_nhwc(input_model.ctx, cams, (int(orig_sz[0]), int(orig_sz[1])))
    cams = cams.astype(np.float64)
    return cams * np.expand_dims(np.expand_dims(is_pass_threshold, axis=1), axis=2)


def resize_stack(stack, size, ctx=None):
    """02_cues/utilities.py:20-40: bilinear resize of every (i, j) map of a (B, C, h, w) stack."""
    from ..misc.imutils import default_context

    ctx = ctx or default_context()
    stack = np.asarray(stack)
    B, C, h, w = stack.shape
    src = np.ascontiguousarray(stack.reshape(B * C, h, w), dtype=np.float32)
    dst = ctx.alloc(B * C * size[0] * size[1] * 4)
    _lib.bilinear_resize(ctx, ctx.to_device(src), B * C, h, w, dst, int(size[0]), int(size[1]))
    return ctx.to_host(dst, (B, C, int(size[0]), int(size[1])), np.float32).astype(np.float64)


# ---- seed (localization cue) generation: 02_cues/utilities.py:183-278 -------------------------------------
def _resolve_and_store(cues, localization, class_inds, indices):
    """Overlap resolution by mask area and the pickle layout SEC/DSRG read (03a_sec-dsrg/model.py:238-246).

    Classes are visited from the largest mask to the smallest; each paints its pixels over what is there,
    so a pixel ends up with the SMALLEST mask that covers it.  `'%d_cues'` is np.where of the one-hot
    result: int64 (3, n) rows (class, row, col); `'%d_labels'` the image's class indices."""
    area = localization.sum(axis=(2, 3))
    order = np.argsort(-area)  # per image, largest mask first (same call as the reference: same tie order)
    B, C, H, W = localization.shape
    label = np.zeros((B, H, W), dtype=np.int64)
    rows = np.arange(B)
    for k in range(C):
        cls = order[:, k]
        mask = localization[rows, cls]
        label = np.where(mask != 0, (cls + 1)[:, None, None], label)
    onehot = np.zeros_like(localization)
    for c in range(C):
        onehot[:, c] = label == c + 1
    for i, x in enumerate(indices):
        cues["%d_labels" % x] = class_inds[i]
        cues["%d_cues" % x] = np.array(np.where(onehot[i]))
    return cues


def get_fgbg_cues(cues, H_fg, H_bg, class_inds, indices, thresh):
    """02_cues/utilities.py:183-235: channel 0 = background (3x3-median-filtered summed background activation
    below its 10th-percentile value), channels 1.. = foreground maps above thresh x (max over the BATCH of
    that class -- SURVEY.md Q7)."""
    import scipy.ndimage

    B, C, H, W = H_fg.shape
    loc = np.zeros((B, C + 1, H, W), dtype="int64")
    for b in range(B):
        grad = scipy.ndimage.median_filter(np.sum(H_bg[b], axis=0), 3)
        thr = np.sort(grad.ravel())[int(0.1 * H * W)]
        loc[b, 0] = grad < thr
    for c in range(C):
        loc[:, c + 1] = H_fg[:, c] > thresh * np.max(H_fg[:, c])
    return _resolve_and_store(cues, loc, class_inds, indices)


def get_fg_cues(cues, H_fg, class_inds, indices, thresh):
    """02_cues/utilities.py:237-278: foreground-only variant (no background channel)."""
    B, C, H, W = H_fg.shape
    loc = np.zeros((B, C, H, W), dtype="int64")
    for c in range(C):
        loc[:, c] = H_fg[:, c] > thresh * np.max(H_fg[:, c])
    return _resolve_and_store(cues, loc, class_inds, indices)


def update_cues_adp(cues, gradcam, class_inds, indices, thresh):
    """02_cues/adp_cues.py:304-339 (ADP seed generation): like get_fg_cues, but a class's threshold is
    thresh x its max over THAT image's map (np.max(gradcam, axis=(2, 3))), not over the batch;
    `gradcam` already holds the background / other channels modify_by_htt inserted."""
    gradcam = np.asarray(gradcam)
    loc = (gradcam > thresh * np.max(gradcam, axis=(2, 3))[:, :, None, None]).astype("int64")
    return _resolve_and_store(cues, loc, class_inds, indices)
