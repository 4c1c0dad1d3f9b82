"""One conv layer through wsc_conv2d_nchw, kernel time from the library's per-launch HIP events.

    python profiles/conv_one.py <N> <Cin> <H> <W> <Cout> <k> <stride> <pad> [precision=f16x3] [res=0] [reps=5]
With the A/B build (ab_tmp/libwsscam_ab.so copied over the package's library) WSC_CONV_DEBUG / WSC_CONV_TILE apply.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))
from wsscam import _lib  # noqa: E402


def main():
    a = sys.argv[1:]
    N, Cin, H, W, Cout, k, stride, pad = (int(v) for v in a[:8])
    prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[a[8] if len(a) > 8 else "f16x3"]
    use_res = len(a) > 9 and int(a[9]) != 0
    reps = int(a[10]) if len(a) > 10 else 5
    rng = np.random.default_rng(0)
    ctx = _lib.Context(0)
    for kv in filter(None, os.environ.get("WSC_BENCH_OPT", "").split(",")):  # A/B: path selectors, e.g. 7=0 (no LDS window)
        ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    x = ctx.to_device(rng.normal(0, 1, (N, Cin, H, W)).astype(np.float32))
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = ctx.to_device(rng.normal(0, 1, (N, Cout, Ho, Wo)).astype(np.float32)) if use_res else None
    y = ctx.alloc(N * Cout * Ho * Wo * 4)
    ablation = os.environ.get("WSC_CONV_DEBUG", "0") not in ("", "0")  # (timing-only: results are wrong and may trip the range guard)
    _lib.conv2d_nchw(ctx, x, N, Cin, H, W, w, stride, pad, None, None, res, True, prec, y)
    if ablation:
        ctx.range_status(clear=True)
    else:
        ctx.sync()
    ctx.profile_begin()
    for _ in range(reps):
        _lib.conv2d_nchw(ctx, x, N, Cin, H, W, w, stride, pad, None, None, res, True, prec, y)
    if ablation:
        ctx.range_status(clear=True)
    prof = ctx.profile_end()
    fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
    tot = 0.0
    for name, (calls, ms, work) in prof.items():
        if name.startswith("conv_igemm"):
            tot += ms / reps
            print("  %-40s %d launches, %.1f us per layer" % (name, calls // reps, ms / reps * 1e3))
    print("layer %s: %.1f us, %.0f TFLOP/s algorithmic  [WSC_CONV_DEBUG=%s WSC_CONV_TILE=%s]" % (
        " ".join(a[:8]), tot * 1e3, fl / (tot * 1e-3) / 1e12, os.environ.get("WSC_CONV_DEBUG", ""), os.environ.get("WSC_CONV_TILE", "")))


if __name__ == "__main__":
    main()
