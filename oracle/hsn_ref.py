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


def cv2_resize_u8(img, dsize_wh):
    """cv2.resize(uint8 HWC image, (w, h)) with INTER_LINEAR as OpenCV computes it for 8-bit images (read_batch of
    02_cues/utilities.py:172-176, 03c_hsn/utilities.py:176-181): OpenCV's published fixed-point algorithm
    (modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear<uchar,int,short> + VResizeLinear<uchar,...>,
    INTER_RESIZE_COEF_BITS = 11), restated as its two passes with explicit loops.  cv2 is not in this image: unpinned
    against cv2 itself; the product's numpy / HIP versions are checked against THIS statement bit for bit."""
    import math
    import struct

    def f32(x):
        return struct.unpack("f", struct.pack("f", x))[0]

    def rne(x):  # cvRound: nearest, ties to even
        r = math.floor(x)
        d = x - r
        if d > 0.5 or (d == 0.5 and r % 2 == 1):
            r += 1
        return int(r)

    img = np.asarray(img, dtype=np.uint8)
    sh, sw = img.shape[:2]
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if (sh, sw) == (dh, dw):
        return img.copy()
    src = img.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:  # INTER_LINEAR with scale exactly 2 x 2 is run as INTER_AREA (fast)
        out = np.zeros((dh, dw, 3), np.uint8)
        for y in range(dh):
            for x in range(dw):
                out[y, x] = (src[2 * y, 2 * x] + src[2 * y, 2 * x + 1] + src[2 * y + 1, 2 * x] + src[2 * y + 1, 2 * x + 1] + 2) >> 2
        return out
    scale_x, scale_y = 1.0 / (dw / sw), 1.0 / (dh / sh)
    xofs, ialpha = [], []
    for dx in range(dw):
        fx = f32((dx + 0.5) * scale_x - 0.5)
        sx = math.floor(fx)
        fx = f32(fx - sx)
        if sx < 0:
            fx, sx = 0.0, 0
        if sx >= sw - 1:
            fx, sx = 0.0, sw - 1
        xofs.append(sx)
        ialpha.append((rne(f32(f32(1.0 - fx) * 2048.0)), rne(f32(fx * 2048.0))))
    out = np.zeros((dh, dw, 3), np.uint8)
    for dy in range(dh):
        fy = f32((dy + 0.5) * scale_y - 0.5)
        sy = math.floor(fy)
        fy = f32(fy - sy)
        b0, b1 = rne(f32(f32(1.0 - fy) * 2048.0)), rne(f32(fy * 2048.0))
        r0, r1 = min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)
        for dx in range(dw):
            sx = xofs[dx]
            a0, a1 = ialpha[dx]
            sx1 = min(sx + 1, sw - 1)
            S0 = src[r0, sx] * a0 + src[r0, sx1] * a1
            S1 = src[r1, sx] * a0 + src[r1, sx1] * a1
            v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
            out[dy, dx] = np.clip(v, 0, 255)
    return out


def read_batch_u8(images, size_hw):
    """The uint8 batch of read_batch (02_cues/utilities.py:172-176; 03c_hsn/utilities.py:170-181)."""
    return np.stack([cv2_resize_u8(im, (size_hw[1], size_hw[0])) for im in images])


# ---- ADP class bookkeeping (03c_hsn/adp_cues.py:20-58), restated here so the oracle chain owns its tables ----------------------
ADP_MORPH = ["E.M.S", "E.M.U", "E.M.O", "E.T.S", "E.T.U", "E.T.O", "E.P", "C.D.I", "C.D.R", "C.L", "H.E", "H.K", "H.Y",
             "S.M.C", "S.M.S", "S.E", "S.C.H", "S.R", "A.W", "A.B", "A.M", "M.M", "M.K", "N.P", "N.R.B", "N.R.A", "N.G.M",
             "N.G.W"]
ADP_FUNC = ["G.O", "G.N", "T"]


def adp_class_tables(all_classes=None):
    """(classes, classinds) dicts of ADPCues.__init__ for the 31-class models."""
    classes = {"all": list(all_classes) if all_classes is not None else ADP_MORPH + ADP_FUNC, "morph": ADP_MORPH,
               "func": ADP_FUNC, "valid_morph": ["Background"] + ADP_MORPH, "valid_func": ["Background", "Other"] + ADP_FUNC}
    inds = {}
    for htt in ("morph", "func"):
        inds[htt + "2valid"] = [i for i, x in enumerate(classes["valid_" + htt]) if x in classes[htt]]
        inds["all2" + htt] = [i for i, x in enumerate(classes["all"]) if x in classes["valid_" + htt]]
    return classes, inds
