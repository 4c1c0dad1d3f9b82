"""ResNet50-CAM forward of 32 images (64 samples): one call on one stream against G concurrent calls of 32 / G images on G
contexts (streams) driven by G host threads.  The blocks of ONE launch move through load / MFMA / store phases together
(stores add linearly to a layer's time); independent sub-batches on separate streams are phase-shifted against each other."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wsss-analysis_amd"))
from wsscam import _lib, synth

B, S, C = 32, 321, 20
x_host = synth.image_batch(B, S, 0)[0]
c0 = _lib.Context(0)
sd = synth.resnet50_cam_state_dict(C, seed=0)
net = _lib.Net(c0, _lib.ARCH_RESNET50_CAM, sd, C, _lib.PREC_F16)
h = net.cam_size(S)

def make(ctx, lo, hi):
    n = hi - lo
    return ctx.to_device(np.ascontiguousarray(x_host[lo:hi])), ctx.alloc(n * C * h * h * 4), n

def run(ctx, pack, reps):
    x, cam, n = pack
    for _ in range(reps):
        net.forward_cam(x, n, S, cam, None, ctx=ctx)
    ctx.sync()

full = make(c0, 0, B)
run(c0, full, 3)
R = 10
cam_full = c0.to_host(full[1], (B, C, h, h), np.float32)
for G in (2, 4):
    ctxs = [_lib.Context(0) for _ in range(G)]
    parts = [make(ctxs[g], g * B // G, (g + 1) * B // G) for g in range(G)]
    for g in range(G):
        run(ctxs[g], parts[g], 2)
    for trial in range(3):
        t0 = time.perf_counter(); run(c0, full, R); t_full = (time.perf_counter() - t0) / R * 1e3
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(ctxs[g], parts[g], R)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        t_con = (time.perf_counter() - t0) / R * 1e3
        t0 = time.perf_counter()
        for g in range(G): run(ctxs[g], parts[g], R)
        t_seq = (time.perf_counter() - t0) / R * 1e3
        print("G=%d: one call %.3f ms | %d concurrent calls %.3f ms | one after the other %.3f ms" % (G, t_full, G, t_con, t_seq))
    cam_parts = np.concatenate([ctxs[g].to_host(parts[g][1], (parts[g][2], C, h, h), np.float32) for g in range(G)])
    print("  cams equal:", bool(np.array_equal(cam_full, cam_parts)))
