"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy / scipy restatement of the HistoSegNet post-processing of
03c_hsn/utilities.py (the reference's own numpy code path, cv2.resize replaced by the half-pixel bilinear resize that
tests/test_dataloaders_host.py pins against OpenCV's INTER_LINEAR rule).  Only tests/ may import this.

  grad_cam_post   utilities.py:262-277   (after the einsum)
  modify_by_htt   utilities.py:306-364
  get_cs_gradcam  utilities.py:367-397
  pass_classes / unary   dcrf_process :425, :431
"""
import numpy as np
import scipy.ndimage
import scipy.special


def _resize(m, out_hw):
    H, W = m.shape
    oh, ow = out_hw
    ys = np.clip((np.arange(oh) + 0.5) * H / oh - 0.5, 0, H - 1)
    xs = np.clip((np.arange(ow) + 0.5) * W / ow - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0)[:, None]; wx = (xs - x0)[None, :]
    return (m[y0][:, x0] * (1 - wx) + m[y0][:, x1] * wx) * (1 - wy) + (m[y1][:, x0] * (1 - wx) + m[y1][:, x1] * wx) * wy


def grad_cam_post(cams, conf_scores, is_pass_threshold, orig_sz):
    """cams (B,h,w,C) einsum result -> (B,S,S,C)."""
    old = np.asarray(cams, dtype=np.float64)
    out = np.zeros((old.shape[0], orig_sz[0], orig_sz[1], old.shape[-1]))
    for i in range(out.shape[0]):
        for j in range(out.shape[-1]):
            out[i, :, :, j] = np.maximum(_resize(old[i, :, :, j], (orig_sz[0], orig_sz[1])), 0)
    out = out / np.maximum(np.max(out, axis=(1, 2, 3), keepdims=True), 1e-7)
    return out * np.expand_dims(np.expand_dims(conf_scores * is_pass_threshold, axis=1), axis=2)


def modify_by_htt(gradcam, images, classes, gradcam_adipose=None):
    gradcam = np.array(gradcam, dtype=np.float64)
    func = gradcam_adipose is not None
    exceptions = ["G.O", "G.N", "T"] if func else ["A.W", "A.B", "A.M"]
    bg_ind = classes.index("Background")
    ex_inds = [i for i, c in enumerate(classes) if c in exceptions]
    bg = 0.75 * scipy.special.expit(4 * (np.mean(images, axis=-1) - 240))
    for i in range(bg.shape[0]):
        bg[i] = scipy.ndimage.gaussian_filter(bg[i], sigma=2)
    bg -= np.max(gradcam[:, ex_inds], axis=1)
    gradcam[:, bg_ind] = bg
    if func:
        other_ind = classes.index("Other")
        other = 0.05 * (1 - np.max(gradcam, axis=1))
        gradcam[:, other_ind] = np.max(np.concatenate((other[:, None], gradcam_adipose), axis=1), axis=1)
    return gradcam


def get_cs_gradcam(gradcam, classes, htt_class):
    other_ind = classes.index("Other") if htt_class in ("func", "glas") else -1
    srt = np.sort(gradcam, axis=1)
    maxdiff = srt[:, -1] - srt[:, -2]
    maxind = np.argmax(gradcam, axis=1)
    cs = np.transpose(np.tile(np.expand_dims(maxdiff, axis=-1), gradcam.shape[1]), (0, 3, 1, 2))
    for c in range(gradcam.shape[1]):
        if c != other_ind:
            cs[:, c] *= (maxind == c)
        else:
            cs[:, c] = gradcam[:, c]
    return cs


def pass_classes(probs_i):
    return np.where(np.sum(np.sum(probs_i, axis=1), axis=1) > 0)[0]
