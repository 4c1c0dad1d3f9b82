"""GPU: single conv layers through the production implicit-GEMM kernel (wsc_conv2d_nchw) against
torch fp32 conv2d, over the conv shape classes of the CAM networks.

Tolerances (stated per the north star):
  bf16 mode   -- operands are rounded to bf16, so the oracle is fed the same bf16-rounded operands
                 and the comparison isolates the kernel: |err| <= 2^-8 |ref| (output rounding to
                 bf16) + 2e-3 max|ref| (fp32 accumulation order).
  f16 mode    -- same scheme with IEEE-half rounding: 2^-11 |ref| + 3e-4 max|ref|.
  bf16x3 mode -- split operands, fp32-class: |err| <= 1e-4 max|ref| against the unrounded oracle.
  f16x3 mode  -- split IEEE-half operands (22-bit significand), both planes staged once: |err| <= 4e-6 max|ref|
                 against the unrounded float64 oracle (fp32 accumulation order + the dropped lo*lo term).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import bf16_round, f16_round
from wsscam import _lib

pytestmark = pytest.mark.gpu

# (N, Cin, H, W, Cout, k, stride, pad)
SHAPES = [
    (2, 3, 65, 65, 64, 7, 2, 3),      # ResNet stem (small-Cin path, 7x7 s2)
    (2, 3, 33, 37, 64, 3, 1, 1),      # VGG first layer (small-Cin path, 3x3)
    (3, 64, 21, 21, 64, 1, 1, 0),     # 1x1 64->64
    (2, 64, 21, 23, 256, 1, 1, 0),    # 1x1 64->256
    (2, 256, 17, 17, 64, 1, 1, 0),    # 1x1 256->64
    (2, 64, 19, 19, 64, 3, 1, 1),     # 3x3 s1
    (2, 128, 21, 21, 128, 3, 2, 1),   # 3x3 s2 (odd -> 11)
    (2, 256, 16, 16, 512, 1, 2, 0),   # downsample 1x1 s2
    (1, 512, 9, 9, 512, 3, 1, 1),     # deep 3x3
    (2, 2048, 5, 5, 512, 1, 1, 0),    # K = 2048
    (5, 128, 7, 9, 136, 1, 1, 0),     # Cout not a multiple of the tile (136 -> pad 256), ragged M
]


def _run(ctx, x, w, stride, pad, scale, shift, res, relu, precision, expect_range=False):
    N, Cin, H, W = x.shape
    x_dev = ctx.to_device(x)
    r_dev = ctx.to_device(res) if res is not None else None
    y_dev, shp = _lib.conv2d_nchw(ctx, x_dev, N, Cin, H, W, w, stride, pad, scale, shift, r_dev, relu, precision)
    if not expect_range:
        return ctx.to_host(y_dev, shp, np.float32)
    # a layer that saturates on purpose: the copy completes, reports WSC_ERR_RANGE (the range guard of the IEEE-half modes,
    # include/wsscam.h) with the saturating layer's channel count, and the flag stays up until it is cleared
    out = np.empty(shp, np.float32)
    st = ctx._lib.wsc_memcpy_d2h(ctx.h, out.ctypes.data, y_dev.ptr, out.nbytes)
    assert st == _lib.WSC_ERR_RANGE, st
    assert ctx.range_status(clear=True) == w.shape[0]
    assert ctx.range_status() == 0
    return out


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("precision", [_lib.PREC_BF16, _lib.PREC_F16, _lib.PREC_BF16X3, _lib.PREC_F16X3])
def test_conv_layer(ctx, shape, precision):
    N, Cin, H, W, Cout, k, stride, pad = shape
    rng = np.random.default_rng(abs(hash(shape)) % (2 ** 31))
    x = rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32)
    # asymmetric weights (transpose-detecting): distinct scale per output channel and tap
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    w *= (1.0 + 0.5 * np.arange(Cout, dtype=np.float32) / Cout)[:, None, None, None]
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.normal(0, 0.2, Cout).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = rng.normal(0, 1, (N, Cout, Ho, Wo)).astype(np.float32)

    for use_res, relu in ((False, False), (True, True)):
        y = _run(ctx, x, w, stride, pad, scale, shift, res if use_res else None, relu, precision)
        if precision == _lib.PREC_BF16:
            xr, wr, rr = bf16_round(x), bf16_round(w), bf16_round(res)
        elif precision == _lib.PREC_F16:
            xr, wr, rr = f16_round(x), f16_round(w), f16_round(res)
        else:
            xr, wr, rr = x, w, res
        ref = F.conv2d(torch.from_numpy(xr).double(), torch.from_numpy(wr).double(), stride=stride, padding=pad)
        ref = ref * torch.from_numpy(scale).double()[None, :, None, None] + \
            torch.from_numpy(shift).double()[None, :, None, None]
        if use_res:
            ref = ref + torch.from_numpy(rr).double()
        if relu:
            ref = torch.relu(ref)
        ref = ref.numpy()
        assert y.shape == ref.shape
        mx = np.abs(ref).max()
        err = np.abs(y - ref)
        if precision == _lib.PREC_BF16:
            bound = 2.0 ** -8 * np.abs(ref) + 2e-3 * mx
        elif precision == _lib.PREC_F16:
            bound = 2.0 ** -11 * np.abs(ref) + 3e-4 * mx
        elif precision == _lib.PREC_BF16X3:
            bound = np.full_like(ref, 1e-4 * mx)
        else:
            bound = np.full_like(ref, 4e-6 * mx)
        bad = err > bound
        assert not bad.any(), "shape %s prec %d res %s: %d bad, max err %.4g (max|ref| %.3g) at %s" % (
            shape, precision, use_res, bad.sum(), err.max(), mx, np.unravel_index(err.argmax(), err.shape))


def test_conv_identity_weights_detect_transposes(ctx):
    """A = asymmetric ramp, B = identity-like 1x1 kernel: output must equal the input channel map."""
    N, C, H, W = 1, 64, 8, 8
    x = (np.arange(N * C * H * W, dtype=np.float32).reshape(N, C, H, W) % 251) / 16.0
    w = np.zeros((64, 64, 1, 1), np.float32)
    perm = (np.arange(64) * 7 + 3) % 64
    w[np.arange(64), perm, 0, 0] = 1.0
    y = _run(ctx, x, w, 1, 0, None, None, None, False, _lib.PREC_BF16)
    assert np.array_equal(y, bf16_round(x)[:, perm])


@pytest.mark.parametrize("precision,tol", [(_lib.PREC_BF16X3, 1e-4), (_lib.PREC_F16X3, 4e-6)])
def test_conv_random_shapes_sweep(ctx, precision, tol):
    """Seeded random sweep over shapes the fixed list does not hit: odd sizes, 5x5 / 1x3 kernels, pads that differ
    from k//2, stride-2 1x1 heads (IRNet), Cout = 8 ... 264, batch sizes whose M straddles tile boundaries."""
    rng = np.random.default_rng(2024)
    for it in range(24):
        Cin = int(rng.choice([64, 128, 192]))
        k = int(rng.choice([1, 1, 3, 5]))
        stride = int(rng.choice([1, 2]))
        pad = int(rng.integers(0, k // 2 + 2)) if k > 1 else 0
        H, W = int(rng.integers(max(k, 2), 30)), int(rng.integers(max(k, 2), 30))
        N = int(rng.integers(1, 5))
        Cout = int(rng.choice([8, 32, 40, 64, 72, 128, 200, 264]))
        if (H + 2 * pad - k) // stride + 1 < 1 or (W + 2 * pad - k) // stride + 1 < 1:
            continue
        x = rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32)
        w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        shift = rng.normal(0, 0.2, Cout).astype(np.float32)
        y = _run(ctx, x, w, stride, pad, scale, shift, None, bool(it & 1), precision)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), stride=stride, padding=pad)
        ref = ref * torch.from_numpy(scale).double()[None, :, None, None] + torch.from_numpy(shift).double()[None, :, None, None]
        if it & 1:
            ref = torch.relu(ref)
        ref = ref.numpy()
        assert y.shape == ref.shape, (it, y.shape, ref.shape)
        assert np.abs(y - ref).max() <= tol * max(np.abs(ref).max(), 1e-3), (it, N, Cin, H, W, Cout, k, stride, pad, np.abs(y - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("precision", [_lib.PREC_F16, _lib.PREC_BF16X3, _lib.PREC_F16X3])
def test_conv_square_tile_large_layer(ctx, precision):
    """A layer big enough for the 256 x 256 tile (CoutPad % 256 == 0, >= 4 K-steps, >= 256 tiles: conv_igemm.hip tile
    choice; 776 tiles = 3 whole rounds on the square kernel + a 128 x 128 remainder launch), with a ragged last tile row, residual and ReLU; fp32 oracle (float64 is too slow at 58 GFLOP)."""
    N, Cin, H, W, Cout, k = 50, 64, 63, 63, 256, 3
    rng = np.random.default_rng(77)
    x = rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    w *= (1.0 + 0.5 * np.arange(Cout, dtype=np.float32) / Cout)[:, None, None, None]
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.normal(0, 0.2, Cout).astype(np.float32)
    res = rng.normal(0, 1, (N, Cout, H, W)).astype(np.float32)
    assert ((N * H * W + 255) // 256) * (Cout // 256) >= 768
    y = _run(ctx, x, w, 1, 1, scale, shift, res, True, precision)
    if precision == _lib.PREC_F16:
        xr, wr, rr = f16_round(x), f16_round(w), f16_round(res)
    else:
        xr, wr, rr = x, w, res
    ref = F.conv2d(torch.from_numpy(xr), torch.from_numpy(wr), stride=1, padding=1)
    ref = torch.relu(ref * torch.from_numpy(scale)[None, :, None, None] + torch.from_numpy(shift)[None, :, None, None] +
                     torch.from_numpy(rr)).numpy()
    mx = np.abs(ref).max()
    err = np.abs(y - ref)
    # (f16x3: the fp32 oracle's own accumulation error is the larger part of the 1e-5)
    bound = 2.0 ** -11 * np.abs(ref) + 3e-4 * mx if precision == _lib.PREC_F16 else np.full_like(ref, 1e-5 * mx if precision == _lib.PREC_F16X3 else 1e-4 * mx)
    assert not (err > bound).any(), "max err %.4g (max|ref| %.3g)" % (err.max(), mx)


@pytest.mark.parametrize("shape", [(2, 3, 65, 65, 64, 7, 2, 3), (3, 64, 21, 23, 256, 1, 1, 0), (2, 256, 17, 17, 64, 1, 1, 0),
                                   (2, 64, 19, 19, 64, 3, 1, 1), (2, 128, 21, 21, 128, 3, 2, 1), (2, 256, 16, 16, 512, 1, 2, 0),
                                   (3, 512, 13, 11, 512, 3, 1, 1), (2, 1024, 9, 9, 256, 1, 1, 0)])
@pytest.mark.parametrize("precision", [_lib.PREC_F16, _lib.PREC_F16X3])
def test_conv_fast_variants_equal_generic_path(ctx, shape, precision):
    """The FAST kernel variants (case-free epilogue with ReLU + saturation as one median, pointwise prologue, one-K-step
    tiles at 128 VGPRs, unrolled K loop) against the generic path of the same kernel (precision | CONV_GENERIC):
    the same fp16 values (0.0 == -0.0), with and without residual / ReLU, including outputs that saturate at 65504."""
    N, Cin, H, W, Cout, k, stride, pad = shape
    rng = np.random.default_rng(Cin * 7 + Cout + k)
    x = rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    scale[:4] = 3.0e4  # a few channels overflow the fp16 range: the saturation path is compared too
    shift = rng.normal(0, 0.2, Cout).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = rng.normal(0, 1, (N, Cout, Ho, Wo)).astype(np.float32)
    for use_res, relu in [(False, True), (True, True), (True, False)]:
        # (both epilogues must also raise the range flag: the generic one and the FAST one)
        y_ref = _run(ctx, x, w, stride, pad, scale, shift, res if use_res else None, relu, precision | _lib.CONV_GENERIC, expect_range=True)
        y = _run(ctx, x, w, stride, pad, scale, shift, res if use_res else None, relu, precision, expect_range=True)
        # (f16x3: hi saturates at 65504 and lo adds what is left of the clamped value: 65504 again)
        assert np.isfinite(y).all() and np.abs(y).max() == 65504.0
        assert np.array_equal(y, y_ref), (shape, use_res, relu, np.abs(y - y_ref).max())


@pytest.mark.parametrize("shape", [(3, 128, 41, 41, 128), (2, 256, 21, 21, 256), (5, 64, 7, 9, 128), (1, 64, 40, 33, 128),
                                   (7, 128, 5, 5, 256), (2, 64, 1, 1, 128), (2, 64, 3, 50, 128), (64, 256, 21, 21, 256),
                                   (4, 64, 19, 23, 64), (2, 128, 40, 40, 64)])
def test_conv_lds_window_equals_per_tap_staging(ctx, shape):
    """The LDS input window of the f16x3 3 x 3 / stride 1 layers (conv_igemm.hip, WPT > 0: all nine taps of a 32-channel
    chunk read ONE window of the zero-padded input raster, staged once per chunk) against the per-tap A tiles of the same
    kernel (ctx option OPT_CONV_WINDOW = 0): the same MFMA sequence on the same operands -- identical bits -- with and
    without residual, on tiles that cross image rows, images and the end of the batch, 1 x 1 and 3 x 50 maps, and the
    ResNet50 layer3 shape at the bench's 64 samples; and against the float64 oracle at the f16x3 bound."""
    N, Cin, H, W, Cout = shape
    rng = np.random.default_rng(N * 131 + Cin + H * 7 + W)
    x = rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    w *= (1.0 + 0.5 * np.arange(Cout, dtype=np.float32) / Cout)[:, None, None, None]
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.normal(0, 0.2, Cout).astype(np.float32)
    res = rng.normal(0, 1, (N, Cout, H, W)).astype(np.float32)
    for use_res, relu in [(False, True), (True, False)]:
        with ctx.option(_lib.OPT_CONV_WINDOW, 0):
            y_tap = _run(ctx, x, w, 1, 1, scale, shift, res if use_res else None, relu, _lib.PREC_F16X3)
        y_win = _run(ctx, x, w, 1, 1, scale, shift, res if use_res else None, relu, _lib.PREC_F16X3)
        assert np.array_equal(y_win, y_tap), (shape, use_res, np.abs(y_win - y_tap).max())
    if N <= 8:
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
        ref = ref * scale[None, :, None, None] + shift[None, :, None, None] + res
        assert np.abs(y_win - ref).max() <= 4e-6 * np.abs(ref).max()  # the f16x3 bound of this file


def test_context_option_restores_the_previous_value(ctx):
    """Context.option(): the block ends with the value the selector HAD, not with the library default -- an
    earlier set_option or an enclosing option() block stays in force, so an A/B test cannot silently compare default vs default."""
    o = _lib.OPT_CONV_WINDOW
    assert ctx.get_option(o) == _lib.OPT_DEFAULTS[o] == 1
    ctx.set_option(o, 0)
    try:
        with ctx.option(o, 1):
            assert ctx.get_option(o) == 1
            with ctx.option(o, 0):
                assert ctx.get_option(o) == 0
            assert ctx.get_option(o) == 1
        assert ctx.get_option(o) == 0
    finally:
        ctx.set_option(o, _lib.OPT_DEFAULTS[o])
