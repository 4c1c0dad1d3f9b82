"""Mean-field loop alone (32 images, 321 x 321, M = 21, T = 10, labels only): wall time per call and the library's per-class
kernel times, for the environment as it is.  Used by crf_ab.sh for A/B runs of env switches / rebuilt variants."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wsss-analysis_amd"))
from wsscam import _lib, synth

B, H, W, M = int(os.environ.get("AB_B", 32)), 321, 321, int(os.environ.get("AB_M", 21))
cfg = (1.5, 3.0, 40.0, 13.0, 10.0, 10)
rng = np.random.default_rng(7)
imgs = np.stack([synth.synth_image(rng, H, W) for _ in range(B)])
U = (-np.log(np.clip(rng.dirichlet(np.ones(M) * 0.3, size=(B, H * W)).transpose(0, 2, 1), 1e-5, 1))).astype(np.float32)
ctx = _lib.Context(0)
for kv in os.environ.get("AB_OPTS", "").split(","):  # e.g. AB_OPTS=9:0 -> ctx.set_option(9, 0)
    if kv:
        o, v = kv.split(":")
        ctx.set_option(int(o), int(v))
rgb = ctx.to_device(imgs)
u = ctx.to_device(U)
a = ctx.alloc(B * H * W * 4)
_lib.Crf(ctx, rgb, B, H, W, cfg[0], cfg[2], cfg[3]).close()  # (a size's tile vertex sets -- the on-chip Gaussian message -- come with its second use)
crf = _lib.Crf(ctx, rgb, B, H, W, cfg[0], cfg[2], cfg[3])
for _ in range(3):
    crf.inference(u, M, cfg[1], cfg[4], cfg[5], None, a)
ctx.sync()
R = int(os.environ.get("AB_R", 10))
best = 1e9
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(R):
        crf.inference(u, M, cfg[1], cfg[4], cfg[5], None, a)
    ctx.sync()
    best = min(best, (time.perf_counter() - t0) / R * 1e3)
ctx.profile_begin()
for _ in range(3):
    crf.inference(u, M, cfg[1], cfg[4], cfg[5], None, a)
ctx.sync()
prof = ctx.profile_end()
lab = ctx.to_host(a, (B, H * W), np.int32)
print("loop %.3f ms/call  on_chip=%s  label checksum %d" % (best, crf.gaussian_on_chip(M), int(lab.astype(np.int64).sum())))
for k, (calls, ms, work) in prof.items():
    if calls:
        print("    %-28s %4d launches  %8.1f us avg  %7.3f ms/call" % (k, calls // 3, ms / calls * 1e3, ms / 3))
