"""misc.imutils helpers used by the CAM path.  The reference imports them from a `misc`
package that is git-ignored there (03b_irn/.gitignore); semantics follow upstream
jiwoon-ahn/irn as recorded in SURVEY.md Appendix A.  Call sites:
03b_irn/step/make_cam.py:41-42, 03b_irn/step/cam_to_ir_label.py:35,47,52,67."""
import numpy as np

from .. import _lib


def get_strided_size(orig_size, stride):
    return ((int(orig_size[0]) - 1) // stride + 1, (int(orig_size[1]) - 1) // stride + 1)


def get_strided_up_size(orig_size, stride):
    strided_size = get_strided_size(orig_size, stride)
    return strided_size[0] * stride, strided_size[1] * stride


def pil_resize(img, size, order):
    """misc/imutils.py of the upstream IRN code base (the module is not vendored in the reference tree; its MSF datasets
    call it at voc12/dataloader.py:236, adp/dataloader.py:228, deepglobe/dataloader.py:206): PIL resize of a uint8 HWC
    image to size = (height, width), bicubic for order 3, nearest for order 0."""
    from PIL import Image

    if size[0] == img.shape[0] and size[1] == img.shape[1]:
        return img
    if order == 3:
        resample = Image.BICUBIC
    elif order == 0:
        resample = Image.NEAREST
    else:
        raise ValueError("pil_resize: order %r (3 = bicubic, 0 = nearest)" % (order,))
    return np.asarray(Image.fromarray(np.uint8(img)).resize(size[::-1], resample))


def pil_rescale(img, scale, order):
    """The multi-scale inference rescale (args.cam_scales): target = round(H * scale) x round(W * scale)."""
    height, width = img.shape[:2]
    target_size = (int(np.round(height * scale)), int(np.round(width * scale)))
    return pil_resize(img, target_size, order)


def scale_images(img_u8, scales):
    """[img at every scale of `scales`] as the MSF datasets build them (s == 1: the image itself)."""
    return [img_u8 if s == 1 else pil_rescale(img_u8, s, order=3) for s in scales]


def HWC_to_CHW(img):
    return np.transpose(img, (2, 0, 1))


def unary_from_labels(labels, n_labels, gt_prob, zero_unsure=True):
    """pydensecrf.utils.unary_from_labels (third-party; restated from its published behaviour)."""
    assert 0 < gt_prob < 1, "`gt_prob must be in (0,1)."
    labels = np.asarray(labels).flatten()
    n_energy = -np.log((1.0 - gt_prob) / (n_labels - 1))
    p_energy = -np.log(gt_prob)
    U = np.full((n_labels, len(labels)), n_energy, dtype="float32")
    U[labels - 1 if zero_unsure else labels, np.arange(U.shape[1])] = p_energy
    if zero_unsure:
        U[:, labels == 0] = -np.log(1.0 / n_labels)
    return U


def unary_from_softmax(sm, scale=None, clip=1e-5):
    """pydensecrf.utils.unary_from_softmax, scale=None branch (03c_hsn/utilities.py:431)."""
    num_cls = sm.shape[0]
    if scale is not None:
        assert 0 < scale <= 1, "`scale` needs to be in (0,1]"
        uniform = np.ones(sm.shape) / num_cls
        sm = scale * sm + (1 - scale) * uniform
    if clip is not None:
        sm = np.clip(sm, clip, 1.0)
    return -np.log(sm).reshape([num_cls, -1]).astype(np.float32)


_ctx_cache = {}


def default_context(device=0):
    if device not in _ctx_cache:
        _ctx_cache[device] = _lib.Context(device)
    return _ctx_cache[device]


def crf_inference_label(img, labels, dataset=None, t=10, n_labels=21, gt_prob=0.7, ctx=None):
    """imutils.crf_inference_label(img, labels, dataset, n_labels=...) -> (H, W) label map.

    upstream irn parameters: unary_from_labels(gt_prob=0.7, zero_unsure=False);
    addPairwiseGaussian(sxy=3, compat=3); addPairwiseBilateral(sxy=50, srgb=5, compat=10).
    The fork's extra positional `dataset` argument is accepted and ignored (its per-dataset
    parameters are not in the reference tree)."""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
    h, w = img.shape[:2]
    U = np.ascontiguousarray(unary_from_labels(labels, n_labels, gt_prob=gt_prob, zero_unsure=False))
    rgb_dev = ctx.to_device(img)
    u_dev = ctx.to_device(U)
    am_dev = ctx.alloc(h * w * 4)
    crf = _lib.Crf(ctx, rgb_dev, 1, h, w, 3.0, 50.0, 5.0)
    crf.inference(u_dev, n_labels, 3.0, 10.0, t, None, am_dev)
    out = ctx.to_host(am_dev, (h, w), np.int32)
    crf.close()
    return out.astype(np.int64)


def crf_inference(img, crf_config, num_classes, featmap, use_log=True, ctx=None):
    """lib.crf.crf_inference of 03a_sec-dsrg (call sites SEC.py:275, DSRG.py:328, model.py:689-693; the file
    itself is not in the reference tree -- upstream SEC/DSRG semantics): unary = -log(featmap) (or -featmap),
    Gaussian + bilateral pairwise terms from crf_config {g_sxy, g_compat, bi_sxy, bi_srgb, bi_compat,
    iterations}; returns the marginals as (H, W, C) float32."""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
    h, w = img.shape[:2]
    fm = np.asarray(featmap, dtype=np.float32).reshape(h, w, num_classes)
    U = -np.log(fm) if use_log else -fm
    U = np.ascontiguousarray(np.transpose(U, (2, 0, 1)).reshape(num_classes, -1), dtype=np.float32)
    rgb_dev = ctx.to_device(img)
    u_dev = ctx.to_device(U)
    q_dev = ctx.alloc(num_classes * h * w * 4)
    crf = _lib.Crf(ctx, rgb_dev, 1, h, w, crf_config["g_sxy"], crf_config["bi_sxy"], crf_config["bi_srgb"])
    crf.inference(u_dev, num_classes, crf_config["g_compat"], crf_config["bi_compat"], int(crf_config["iterations"]),
                  q_dev, None)
    Q = ctx.to_host(q_dev, (num_classes, h, w), np.float32)
    crf.close()
    return np.ascontiguousarray(np.transpose(Q, (1, 2, 0)))
