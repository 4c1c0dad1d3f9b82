"""HistoSegNet post-processing on the device -- mirror of 03c_hsn/utilities.py.

dcrf_process :399-445 is the only in-tree pydensecrf call site; here the whole batch goes
through libwsscam's permutohedral mean-field in one call per distinct class count."""
import numpy as np

from .. import _lib
from ..misc.imutils import default_context, unary_from_softmax


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
