"""Shared helpers of the parity tests (oracle access lives here and in the tests only)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bf16_round(a):
    """Round float32 array to bfloat16 (RNE) and back."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def f16_round(a):
    """Round float32 array to IEEE half (RNE, saturating like the kernels) and back."""
    a = np.clip(np.ascontiguousarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).astype(np.float32)


_crf_lib = None


def crf_oracle_lib():
    global _crf_lib
    if _crf_lib is None:
        import __graft_entry__ as ge

        path = ge.build_oracle()
        lib = ctypes.CDLL(path)
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        lib.densecrf_ref_inference.argtypes = [u8p, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int] + \
            [ctypes.c_float] * 5 + [ctypes.c_int, f32p, i32p, i32p]
        lib.densecrf_ref_inference.restype = ctypes.c_int
        lib.densecrf_ref_lattice_filter.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, f32p, ctypes.c_int]
        lib.densecrf_ref_lattice_filter.restype = ctypes.c_int
        _crf_lib = lib
    return _crf_lib


def crf_oracle(rgb, U, cfg):
    """C restatement of pydensecrf's DenseCRF2D inference: returns (Q (M,N), argmax (N,), [V_g, V_b])."""
    lib = crf_oracle_lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    U = np.ascontiguousarray(U, dtype=np.float32)
    H, W, _ = rgb.shape
    M = U.shape[0]
    q = np.empty((M, H * W), np.float32)
    am = np.empty(H * W, np.int32)
    ls = np.zeros(2, np.int32)
    lib.densecrf_ref_inference(rgb, H, W, U, M, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], int(cfg[5]), q, am, ls)
    return q, am, ls


def crf_exact(rgb, U, cfg):
    """Exact O(N^2) mean-field with true Gaussian kernels (float64) -- the model the lattice approximates."""
    H, W, _ = rgb.shape
    M = U.shape[0]
    N = H * W
    yy, xx = np.mgrid[0:H, 0:W]
    fg = np.stack([xx.ravel() / cfg[0], yy.ravel() / cfg[0]], 1).astype(np.float64)
    fb = np.concatenate([np.stack([xx.ravel() / cfg[2], yy.ravel() / cfg[2]], 1),
                         rgb.reshape(N, 3).astype(np.float64) / cfg[3]], 1)

    def K(f):
        d = ((f[:, None, :] - f[None, :, :]) ** 2).sum(-1)
        return np.exp(-0.5 * d)

    KG, KB = K(fg), K(fb)
    nG = 1 / np.sqrt(KG.sum(1) + 1e-20)
    nB = 1 / np.sqrt(KB.sum(1) + 1e-20)

    def sm(E):
        E = E - E.max(0, keepdims=True)
        e = np.exp(E)
        return e / e.sum(0, keepdims=True)

    U = U.astype(np.float64)
    Q = sm(-U)
    for _ in range(int(cfg[5])):
        mG = nG[None, :] * ((Q * nG[None, :]) @ KG.T)
        mB = nB[None, :] * ((Q * nB[None, :]) @ KB.T)
        Q = sm(-U + cfg[1] * mG + cfg[4] * mB)
    return Q


def synth_crf_case(rng, H, W, M, sharp=3.0):
    """Blobby image + soft class probabilities that follow (but do not equal) the image regions."""
    from oracle import cnn_ref

    rgb = cnn_ref.synth_image(rng, H, W)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    logits = rng.normal(0, 0.6, (M, H, W)).astype(np.float32)
    for m in range(M):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        r = rng.uniform(0.15, 0.45) * min(H, W)
        logits[m] += sharp * np.exp(-(((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r)))
    p = np.exp(logits - logits.max(0, keepdims=True))
    p /= p.sum(0, keepdims=True)
    U = -np.log(np.clip(p, 1e-5, 1.0)).reshape(M, -1).astype(np.float32)
    return rgb, np.ascontiguousarray(U), p


# ---- chain-level parity on BASELINE config 3 (CNN -> unaries -> dense CRF -> labels) ---------------------------------
def oracle_chain(x_pair, rgb, sd, cfg, num_classes=20, bg=0.15, cam=None):
    """The reference chain for ONE image, all fp32 on the CPU: resnet50_cam.CAM.forward (net/resnet50_cam.py:55-70) ->
    make_cam's upsample + per-class max-normalisation for ALL classes at the image size (step/make_cam.py:64-76) ->
    [bg | maps] probabilities -> -log(clip(p)) unaries (eval_cam.py:49-51, 03c_hsn/utilities.py:431) -> DenseCRF
    (oracle/densecrf_ref.c).  Returns (labels (H*W,) int32, unaries (M, H*W), cam)."""
    import torch

    from oracle import cnn_ref

    H, W, _ = rgb.shape
    if cam is None:
        with torch.no_grad():
            cam = cnn_ref.resnet50_cam_forward(torch.from_numpy(np.ascontiguousarray(x_pair)), sd)
    with torch.no_grad():
        _, hi = cnn_ref.make_cam_tail(cam, (H, W), torch.arange(num_classes))
    v = np.concatenate([np.full((1, H * W), bg, np.float32), hi.numpy().reshape(num_classes, -1)], 0)
    U = np.ascontiguousarray(-np.log(np.clip(v / v.sum(0, keepdims=True), 1e-5, 1.0)).astype(np.float32))
    _, lab, _ = crf_oracle(rgb, U, cfg)
    return lab, U, cam


def product_chain(ctx, net, x, rgb, cfg, num_classes=20, bg=0.15):
    """The product chain for a batch through the C ABI: wsc_net_forward_cam -> wsc_cam_unary_pm -> wsc_crf_create ->
    wsc_crf_inference_pm (labels only).  x float32 [B][2][3][S][S], rgb uint8 [B][S][S][3] -> labels int32 [B][S*S]."""
    from wsscam import _lib

    B, S = x.shape[0], x.shape[-1]
    h = net.cam_size(S)
    Mp = (num_classes + 1 + 3) // 4 * 4
    x_dev, rgb_dev = ctx.to_device(np.ascontiguousarray(x)), ctx.to_device(np.ascontiguousarray(rgb))
    cam_dev = ctx.alloc(B * num_classes * h * h * 4)
    u_dev, l_dev = ctx.alloc(B * Mp * S * S * 4), ctx.alloc(B * S * S * 4)
    net.forward_cam(x_dev, B, S, cam_dev, None)
    _lib.cam_unary(ctx, cam_dev, B, num_classes, h, h, S, S, bg, u_dev, pixel_major=True)
    crf = _lib.Crf(ctx, rgb_dev, B, S, S, cfg[0], cfg[2], cfg[3])
    crf.inference(u_dev, num_classes + 1, cfg[1], cfg[4], int(cfg[5]), None, l_dev, pixel_major=True)
    lab = ctx.to_host(l_dev, (B, S * S), np.int32)
    crf.close()
    for d in (x_dev, rgb_dev, cam_dev, u_dev, l_dev):
        d.free()
    return lab


def label_parity(lab, ref, n_class):
    """Per-image label agreement and the mIoU of `lab` scored against `ref` as ground truth, the way eval_cam scores
    pseudo-labels (03b_irn/step/eval_cam.py:89-115: confusion matrix, IoU = TP / (TP + FP + FN) per class, mean over the
    classes that occur)."""
    lab, ref = np.asarray(lab).reshape(len(ref), -1), np.asarray(ref).reshape(len(ref), -1)
    agree = [(a == b).mean() for a, b in zip(lab, ref)]
    conf = np.bincount((ref.ravel() * n_class + lab.ravel()).astype(np.int64), minlength=n_class * n_class).reshape(n_class, n_class)
    tp = np.diag(conf).astype(np.float64)
    denom = conf.sum(0) + conf.sum(1) - tp
    present = denom > 0
    return {"label_agreement_min": float(min(agree)), "label_agreement_mean": float(np.mean(agree)),
            "miou_vs_oracle": float((tp[present] / denom[present]).mean()), "classes_present": int(present.sum()),
            "images": int(len(ref))}


def oracle_chain_hsn_adp(images, sd, alpha, thr, cfgs, all_classes=None, size=None, adipose_as_written=True):
    """The reference chain of 03c_hsn/demo.py:271-380 (ADP) for a list of (S, S, 3) uint8 patches, all fp32 / float64 on the
    CPU: VGG16 features (torch) -> sigmoid scores -> Grad-CAM einsum -> bilinear upsample, ReLU, / max, x score x pass ->
    per HTT type: valid-class stack, modify_by_htt (background / other channels, 03c_hsn/utilities.py:306-364),
    get_cs_gradcam (:367-397), dense CRF on the classes with mass (:399-445, oracle/densecrf_ref.c).
    sd: torch state dict; alpha (F, C); thr: scalar threshold; cfgs {'morph', 'func'} 6-vectors; size: the network size the
    patches are resized to first (ADPCues.read_batch, adp_cues.py:122-128: cv2.resize into a uint8 batch -- the oracle's own
    loop statement of OpenCV's 8-bit rule); None = the patches are at the network size.  The class tables are the oracle's
    own (oracle/hsn_ref.py), not the product's.  adipose_as_written: demo.py:368-369 takes the positions of A.W / A.B / A.M in
    classes['morph'] and indexes the VALID morph stack with them (channel 0 = Background: the maps picked are S.R, A.W, A.B);
    False = the evidently intended channels, only to show that a test can tell the two apart.
    Returns {'morph': [label maps], 'func': [label maps]}."""
    import scipy.ndimage
    import scipy.special
    import torch

    from oracle import cnn_ref, hsn_ref

    raw = np.stack(images) if size is None else hsn_ref.read_batch_u8(images, (size, size))
    images = list(raw)
    n, S = raw.shape[0], raw.shape[1]
    x = (raw - 193.09203) / 56.450138
    xt = torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).astype(np.float32).copy())
    with torch.no_grad():
        feat = cnn_ref.plain_features(xt, sd, "vgg16", cnn_ref.VGG16_CFG)
        sc = torch.sigmoid(torch.nn.functional.linear(feat.mean((2, 3)), sd["vgg16.classifier.0.weight"],
                                                      sd["vgg16.classifier.0.bias"])).numpy().astype(np.float64)
    cams = np.einsum("ijkl,lm->ijkm", np.transpose(feat.numpy(), (0, 2, 3, 1)).astype(np.float64), alpha)
    up = np.maximum(torch.nn.functional.interpolate(torch.from_numpy(np.transpose(cams, (0, 3, 1, 2))), (S, S),
                                                    mode="bilinear", align_corners=False).numpy(), 0)
    H = up / np.maximum(up.max(axis=(1, 2, 3), keepdims=True), 1e-7) * (sc * (sc >= thr))[:, :, None, None]
    a_classes, a_inds = hsn_ref.adp_class_tables(all_classes)
    Y, out = {}, {"morph": [], "func": []}
    for htt in ("morph", "func"):
        valid = a_classes["valid_" + htt]
        Y[htt] = np.zeros((n, len(valid), S, S))
        Y[htt][:, a_inds[htt + "2valid"]] = H[:, a_inds["all2" + htt]]
        bgm = np.stack([scipy.ndimage.gaussian_filter(0.75 * scipy.special.expit(4 * (raw[i].mean(-1) - 240)), sigma=2)
                        for i in range(n)])
        if htt == "morph":
            Y[htt][:, 0] = bgm - Y[htt][:, [valid.index(c) for c in ("A.W", "A.B", "A.M")]].max(1)
        else:
            Y[htt][:, 0] = bgm - Y[htt][:, [valid.index(c) for c in ("G.O", "G.N", "T")]].max(1)
            other = 0.05 * (1 - Y[htt].max(1))
            adi = Y["morph"][:, [(a_classes["morph"] if adipose_as_written else a_classes["valid_morph"]).index(c)
                                 for c in ("A.W", "A.B", "A.M")]]
            Y[htt][:, 1] = np.maximum(other, adi.max(1))
        srt = np.sort(Y[htt], axis=1)
        cs = (srt[:, -1] - srt[:, -2])[:, None] * (np.arange(len(valid))[None, :, None, None] == Y[htt].argmax(1)[:, None])
        if htt == "func":
            cs[:, 1] = Y[htt][:, 1]
        for b in range(n):
            keep = np.where(cs[b].sum(axis=(1, 2)) > 0)[0]
            U = np.ascontiguousarray(-np.log(np.clip(cs[b][keep], 1e-5, 1.0)).reshape(len(keep), -1).astype(np.float32))
            _, ar, _ = crf_oracle(images[b], U, tuple(cfgs[htt]))
            out[htt].append(keep[ar.reshape(S, S)])
    return out
