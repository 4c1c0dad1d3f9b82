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
