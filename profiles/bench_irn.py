"""BASELINE config 4 (IRNet inference) stage timings on one MI355X -- not the contract bench (bench.py).

    python profiles/bench_irn.py [--batch 8] [--arch vgg16|resnet50] [--precision f16]
Per image: EdgeDisplacement on the [orig, flip] pair zero-padded to 512x512 (make_sem_seg_labels.py:46), then the
random walk of K = 2 strided CAMs at 94 x 125 (375x500 VOC image), beta 10, 2^8 steps, then the x4 upsample.
Prints one JSON line: ms per image of each stage, images/s, and the work the reference's dense formulation
(8 squarings of the 11750 x 11750 transition matrix) would have needed."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wsss-analysis_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--arch", default="vgg16", choices=["vgg16", "resnet50"])
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16", "bf16", "bf16x3"])
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np

    from wsscam import _lib, synth
    from wsscam.misc.indexing import PathIndex

    ctx = _lib.Context(0)
    if os.environ.get("BENCH_IRN_TILED"):  # A/B: 1 = the tiled random-walk step whatever the batch, 0 = the flat one
        ctx.set_option(_lib.OPT_RW_TILED, int(os.environ["BENCH_IRN_TILED"]))
    prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[args.precision]
    sd = synth.irn_state_dict(args.arch, 0)  # seeded random weights (no trained IRNet weights offline)
    arch = _lib.ARCH_VGG16_IRN if args.arch == "vgg16" else _lib.ARCH_RESNET50_IRN
    net = _lib.Net(ctx, arch, sd, 20, prec)
    B, S, fh, fw = args.batch, 512, 81, 81  # 321x321 network input -> (321-1)//4+1 = 81
    rng = np.random.default_rng(0)
    x = np.zeros((B, 2, 3, S, S), np.float32)
    x[..., :321, :321] = rng.normal(0, 1, (B, 2, 3, 321, 321)).astype(np.float32)
    x_dev = ctx.to_device(x)
    edge_dev, dp_dev = ctx.alloc(B * fh * fw * 4), ctx.alloc(B * 2 * fh * fw * 4)
    h, w, K = 94, 125, 2
    dirs, start, yx = PathIndex(5).device_tables()
    cams_dev = ctx.to_device(rng.random((K, h, w)).astype(np.float32))
    e2_dev = ctx.to_device((rng.random((h, w)) ** 2).astype(np.float32))
    rw_dev = ctx.alloc(K * h * w * 4)
    up_dev = ctx.alloc(K * 375 * 500 * 4)

    def timed(fn, reps):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    t_net = timed(lambda: net.forward_edge(x_dev, B, S, fh, fw, edge_dev, dp_dev), args.steps)
    t_rw1 = timed(lambda: _lib.rw_propagate(ctx, cams_dev, e2_dev, K, h, w, dirs, start, yx, 10.0, 256, rw_dev), args.steps)
    RB = int(os.environ.get("BENCH_IRN_RB", 32))  # images per random-walk pass (make_sem_seg_labels mirror: args.irn_batch_images)
    cams_b = ctx.to_device(rng.random((RB, K, h, w)).astype(np.float32))
    e2_b = ctx.to_device((rng.random((RB, h, w)) ** 2).astype(np.float32))
    rw_b = ctx.alloc(RB * K * h * w * 4)
    t_rw = timed(lambda: _lib.rw_propagate_batch(ctx, cams_b, e2_b, [K] * RB, [h] * RB, [w] * RB, dirs, start, yx, 10.0,
                                                 256, rw_b), args.steps) / RB
    t_up = timed(lambda: _lib.bilinear_resize(ctx, rw_dev, K, h, w, up_dev, 375, 500), args.steps)
    per_img = t_net / B + t_rw + t_up
    hw = h * w
    # the driver itself: make_sem_seg_labels.sem_seg_batch on 16 VOC-sized images (host arrays in, label maps out; no file I/O)
    import types

    from wsscam.step import make_sem_seg_labels as mssl

    if args.arch == "vgg16":
        from wsscam.net import vgg16_irn as irn_mod

        model = irn_mod.EdgeDisplacement(None, "voc12", "", 20, None, precision=prec)
    else:
        from wsscam.net import resnet50_irn as irn_mod

        model = irn_mod.EdgeDisplacement(None, 20, precision=prec)
    model.load_state_dict(sd, strict=False)
    model.cuda(0)
    DB = 16
    packs = [{"name": "i%d" % i, "img": rng.normal(0, 1, (2, 3, 375, 500)).astype(np.float32), "size": (375, 500)} for i in range(DB)]
    cam_dicts = [{"keys": np.array([3, 11]), "cam": rng.random((K, h, w)).astype(np.float32)} for _ in range(DB)]
    dargs = types.SimpleNamespace(dataset="voc12", beta=10, exp_times=8, sem_seg_bg_thres=0.25)
    mssl.sem_seg_batch(model, packs, cam_dicts, dargs)
    model.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        mssl.sem_seg_batch(model, packs, cam_dicts, dargs)
    model.ctx.sync()
    t_drv = (time.perf_counter() - t0) / 3 / DB * 1e3
    print(json.dumps({
        "workload": "IRNet inference (BASELINE config 4): %s EdgeDisplacement @512 pad + random walk K=%d %dx%d 2^8 steps"
                    % (args.arch, K, h, w),
        "dtype": args.precision, "batch_images": B,
        "edge_net_ms_per_image": round(t_net / B, 3), "random_walk_ms_per_image": round(t_rw, 3),
        "random_walk_batch_images": RB, "random_walk_ms_single_image_call": round(t_rw1, 3),
        "upsample_ms_per_image": round(t_up, 3), "images_per_s": round(1e3 / per_img, 1),
        "driver_ms_per_image": round(t_drv, 3), "driver_images_per_s": round(1e3 / t_drv, 1), "driver_batch_images": DB,
        "stencil_GFLOP_per_image": round(256 * K * hw * 69 * 2 / 1e9, 2),
        "reference_dense_TFLOP_per_image": round(8 * 2 * hw ** 3 / 1e12, 1)}))


if __name__ == "__main__":
    main()
