"""HistoSegNet post-processing on the device -- mirror of 03c_hsn/utilities.py.

dcrf_process :399-445 is the only in-tree pydensecrf call site; here the whole batch goes
through libwsscam's permutohedral mean-field in one call per distinct class count."""
import numpy as np

from .. import _lib
from ..cues import utilities as cues_utilities
from ..misc.imutils import default_context, unary_from_softmax


def grad_cam(input_model, weights, images, is_pass_threshold, final_layer, conf_scores, orig_sz=[224, 224],
             should_upsample=False):
    """03c_hsn/utilities.py:231-278 -> (B, S, S, C): einsum (no ReLU before the resize), per-map bilinear
    upsample then max(., 0), division by the per-image maximum over all classes, gating by score x pass."""
    cams, _ = cues_utilities.conv_and_cams(input_model, np.asarray(weights), images, relu=False)
    if should_upsample:
        cams = np.maximum(cues_utilities._upsample_nhwc(input_model.ctx, cams, (int(orig_sz[0]), int(orig_sz[1]))), 0)
    cams = cams.astype(np.float64)
    cams = cams / np.maximum(np.max(cams, axis=(1, 2, 3), keepdims=True), 1e-7)
    return cams * np.expand_dims(np.expand_dims(conf_scores * is_pass_threshold, axis=1), axis=2)


def modify_by_htt(gradcam, images, classes, gradcam_adipose=None):
    """03c_hsn/utilities.py:306-364: synthesise the 'Background' (and, for functional types, 'Other')
    channels of an ADP Grad-CAM stack in place and return it.

    background = 0.75 * sigmoid(4 * (mean_rgb - 240)), Gaussian-smoothed (sigma 2), resized to the CAM
    size if needed, minus the strongest exception-class activation; other = max(0.05 * (1 - max_c cam),
    adipose cam)."""
    import scipy.ndimage
    import scipy.special

    func = gradcam_adipose is not None
    exceptions = ["G.O", "G.N", "T"] if func else ["A.W", "A.B", "A.M"]
    bg_ind = classes.index("Background")
    ex_inds = [i for i, c in enumerate(classes) if c in exceptions]
    bg = 0.75 * scipy.special.expit(4 * (np.mean(images, axis=-1) - 240))
    for i in range(bg.shape[0]):
        bg[i] = scipy.ndimage.gaussian_filter(bg[i], sigma=2)
    if bg.shape[1:] != gradcam.shape[2:]:
        bg = cues_utilities.resize_stack(bg[:, None], (gradcam.shape[2], gradcam.shape[3]))[:, 0]
    gradcam[:, bg_ind] = bg - np.max(gradcam[:, ex_inds], axis=1)
    if func:
        other_ind = classes.index("Other")
        other = 0.05 * (1 - np.max(gradcam, axis=1))
        gradcam[:, other_ind] = np.max(np.concatenate((other[:, None], gradcam_adipose), axis=1), axis=1)
    return gradcam


def get_cs_gradcam(gradcam, classes, htt_class):
    """03c_hsn/utilities.py:367-397: class-specific Grad-CAM = (top1 - top2 margin) on the arg-max class,
    zero elsewhere; the functional/glas 'Other' channel passes through unchanged."""
    top2 = np.partition(gradcam, gradcam.shape[1] - 2, axis=1)[:, -2:]
    maxdiff = top2[:, 1] - top2[:, 0]
    maxind = np.argmax(gradcam, axis=1)
    cs = np.zeros_like(gradcam)
    other_ind = classes.index("Other") if htt_class in ("func", "glas") else -1
    for c in range(gradcam.shape[1]):
        cs[:, c] = gradcam[:, c] if c == other_ind else maxdiff * (maxind == c)
    return cs


def dcrf_process(probs, images, config, ctx=None):
    """Run dense CRF, given probability map and input image (03c_hsn/utilities.py:399-445).

    probs  : (B, C, H, W) class probability maps
    images : (B, H, W, 3) original input images
    config : [g_sxy, g_compat, bi_sxy, bi_srgb, bi_compat, n_infer]
    returns (B, H, W) int64 arg-max class map.
    """
    gauss_sxy, gauss_compat, bilat_sxy, bilat_srgb, bilat_compat, n_infer = config
    ctx = ctx or default_context()
    probs = np.asarray(probs)
    num_input_images, num_classes = probs.shape[0], probs.shape[1]
    size = images.shape[1:3]
    H, W = int(size[0]), int(size[1])
    crf = np.zeros((num_input_images, num_classes, H, W))
    # per image: classes with any positive mass (:425)
    pass_inds = [np.where(np.sum(np.sum(probs[i], axis=1), axis=1) > 0)[0] for i in range(num_input_images)]
    rgb = np.ascontiguousarray(np.uint8(images))
    # images sharing a class count M go through the device together
    groups = {}
    for i, p in enumerate(pass_inds):
        if len(p) > 0:
            groups.setdefault(len(p), []).append(i)
    for M, idxs in groups.items():
        Bg = len(idxs)
        U = np.stack([np.ascontiguousarray(unary_from_softmax(probs[i, pass_inds[i]])) for i in idxs])
        rgb_dev = ctx.to_device(rgb[idxs])
        u_dev = ctx.to_device(U.astype(np.float32))
        q_dev = ctx.alloc(Bg * M * H * W * 4)
        d = _lib.Crf(ctx, rgb_dev, Bg, H, W, gauss_sxy, bilat_sxy, bilat_srgb)
        d.inference(u_dev, M, gauss_compat, bilat_compat, int(n_infer), q_dev, None)
        Q = ctx.to_host(q_dev, (Bg, M, H, W), np.float32)
        d.close()
        for j, i in enumerate(idxs):
            crf[i, pass_inds[i]] = Q[j]
    maxconf_crf = np.argmax(crf, axis=1)
    return maxconf_crf
