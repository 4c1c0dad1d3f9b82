"""Does splitting a 32-image mean-field batch into two 16-image halves on two contexts (streams), run concurrently from two
host threads, beat the single 32-image call?  (update blocks take all of a CU's LDS: only LDS-free kernels -- the bilateral
combine / blur passes of the OTHER half -- can share a CU with them.)"""
import sys, threading, time
import numpy as np
sys.path.insert(0, "wsss-analysis_amd")
from wsscam import _lib, synth

B, H, W, M = 32, 321, 321, 21
cfg = (1.5, 3.0, 40.0, 13.0, 10.0, 10)
rng = np.random.default_rng(7)
imgs = np.stack([synth.synth_image(rng, H, W) for _ in range(B)])
U = (-np.log(np.clip(rng.dirichlet(np.ones(M) * 0.3, size=(B, H * W)).transpose(0, 2, 1), 1e-5, 1))).astype(np.float32)

def make(ctx, lo, hi):
    n = hi - lo
    rgb = ctx.to_device(np.ascontiguousarray(imgs[lo:hi]))
    u = ctx.to_device(np.ascontiguousarray(U[lo:hi]))
    a = ctx.alloc(n * H * W * 4)
    crf = _lib.Crf(ctx, rgb, n, H, W, cfg[0], cfg[2], cfg[3])
    return crf, u, a, n

def run(ctx, pack, reps):
    crf, u, a, n = pack
    for _ in range(reps):
        crf.inference(u, M, cfg[1], cfg[4], cfg[5], None, a)
    ctx.sync()

c0, c1, c2 = _lib.Context(0), _lib.Context(0), _lib.Context(0)
full = make(c0, 0, 32)
ha, hb = make(c1, 0, 16), make(c2, 16, 32)
run(c0, full, 2); run(c1, ha, 2); run(c2, hb, 2)
R = 6
for trial in range(3):
    t0 = time.perf_counter(); run(c0, full, R); t_full = (time.perf_counter() - t0) / R * 1e3
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c1, ha, R)), threading.Thread(target=run, args=(c2, hb, R))]
    for t in th: t.start()
    for t in th: t.join()
    t_half = (time.perf_counter() - t0) / R * 1e3
    t0 = time.perf_counter(); run(c1, ha, R); run(c2, hb, R); t_seq = (time.perf_counter() - t0) / R * 1e3
    print("32 images: one call %.3f ms | two 16-image calls concurrently %.3f ms | one after the other %.3f ms" % (t_full, t_half, t_seq))
lab_full = c0.to_host(full[2], (32, H * W), np.int32)
lab_half = np.concatenate([c1.to_host(ha[2], (16, H * W), np.int32), c2.to_host(hb[2], (16, H * W), np.int32)])
print("labels equal:", bool(np.array_equal(lab_full, lab_half)))
