#!/usr/bin/env python
"""bench.py -- images/sec of CAM + dense-CRF pseudo-label generation (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of 32 synthetic VOC-like images resident
in HBM (BASELINE config 2/3: ResNet50 CAM + dense-CRF, 321x321, batch 32 images = 64 samples):

  wsc_net_forward_cam      ResNet50 (strides 2,2,2,1) on orig+flip, 1x1 CAM head, ReLU, flip-add
  wsc_cam_postprocess      make_cam tail for the image's GT classes at its native size
                           (strided /4 map + high-res map, per-class max-normalised)
  wsc_cam_unary            all 20 class maps at 321x321, max-normalised, [bg=0.15 | maps] -> -log(clip(p))
                           unaries, M = 21 (wsc_cam_postprocess + wsc_unary_from_maps fused: same bits)
  wsc_crf_create           Gaussian + bilateral permutohedral lattices of the 32 images (second
                           context/stream: overlaps the conv stack, joined by wsc_ctx_wait)
  wsc_crf_inference        10 mean-field iterations -> arg-max label map

N > 1: one process per GPU (torch.distributed.run), every rank owns its own batch (the dataset is
image-sharded, images[g::G]); no data-path collective, only the timing barrier.  scaling = weak.

The headline (`value`, `dtype` "f16x3") runs the conv stack in the fp32-class mode -- split-half operands and activations,
three MFMA products per term (the reference's arithmetic is fp32); the mean-field loop is fp32 throughout.  Inputs are
resident in HBM.  The same JSON line carries (rank 0, N = 1; --quick skips the extras): `value_steady` (>= 10 s of steps
cycling over four distinct resident batches), `roofline` with the two fractions the north star names (`conv_mfma`,
`crf_hbm`; `frac` = algorithmic rate / peak, `frac_traffic` = measured HBM traffic / peak), `parity` (final label maps of
the product chain against the all-fp32 oracle chain on images of this batch -- computed inside the cpu_baseline leg, the
only place bench.py touches oracle/), and under `stages`: `sum_ms` (sum of the stage times: what the step would take
without overlap), `value_f16` (the fast half-precision mode), `value_M_eq_Kplus1` (SURVEY config 3's other variant:
M = K + 1 per image), `value_end_to_end` (pageable host batch -> pinned staging -> H2D one step ahead on its own stream, D2H of
cam / high_res / labels on their own streams, a finisher thread, .npy files through writer threads).
--scaling strong: the K steps are a FIXED set of K*batch images sharded over the ranks (rank g runs ceil(K/N) steps).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "wsss-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# algorithmic GFLOP per image = 2 samples (orig + flip) through the conv stack incl. the 1x1 head
GFLOP_PER_IMAGE_BY_ARCH = {"resnet50": 54.923, "vgg16": 249.95, "m7": 37.34}
INPUT_SIZE_BY_ARCH = {"resnet50": 321, "vgg16": 321, "m7": 224}
GFLOP_PER_IMAGE = 54.923
S = 321
NUM_CLASSES = 20
CRF_CFG = (1.5, 3.0, 40.0, 13.0, 10.0, 10)  # 03c_hsn/demo.py:157-165 VOC-VGG16 / DeepGlobe
# dense 16-bit MFMA peak (MI355X_MICROARCH.md); the split modes spend three MFMA products per algorithmic multiply-add, so
# the peak of THAT arithmetic is a third of it (the native fp32-input MFMA peak is 157 TFLOP/s)
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0 / 3, "f16x3": 2500.0 / 3}
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per step per GPU")
    ap.add_argument("--precision", default="f16x3", choices=["bf16", "f16", "bf16x3", "f16x3"],
                    help="conv operand mode: f16x3 = fp32-class (headline), f16 / bf16 = fast 16-bit modes, bf16x3 = the round-1 split")
    ap.add_argument("--workload", default="cam_crf", choices=["cam_crf", "cam", "hsn", "make_cam"],
                    help="cam_crf: the BASELINE.json metric; cam: make_cam only; hsn: BASELINE config 5 (HistoSegNet on ADP-like "
                         "321x321 patches: VGG16 Grad-CAM -> modify_by_htt -> cs-gradcam -> dense CRF for the 29 morphological "
                         "and the 5 functional classes, 03c_hsn/demo.py:271-380), an extra measurement")
    ap.add_argument("--arch", default="resnet50", choices=["resnet50", "vgg16", "m7"],
                    help="CAM network (resnet50 is the BASELINE.json configuration; vgg16 / m7 are extra "
                         "measurements of the other conv stacks of the reference, 03b_irn/net/{vgg16,m7}.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="skip the extra measurements (steady-state leg, f16, M=K+1, end-to-end)")
    ap.add_argument("--steady-seconds", type=float, default=10.0, help="length of the value_steady leg (0 = skip)")
    ap.add_argument("--seconds", type=float, default=0.0,
                    help="steady-state run: raise --steps so that the timed region lasts at least this long")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank runs --steps steps on its own batch; strong: --steps x --batch images in total, "
                         "image-sharded over the ranks")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish each step before starting the next (default: the mean-field loop of step i "
                         "overlaps the conv stack + lattice build of step i+1 on separate streams)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the timing barrier (nccl = RCCL; gloo only to exercise the "
                         "N>1 code path on a box with fewer GPUs than ranks, together with WSC_BENCH_DEVICE)")
    return ap.parse_args()


def crf_bytes_per_image(N, M, T, vg, vb):
    """algorithmic HBM bytes of the mean-field loop for one image."""
    per_it = N * M * 4  # read U
    for d, V in ((2, vg), (5, vb)):
        per_it += N * M * 4            # read Q (splat)
        per_it += N * (d + 1) * 8      # splat idx + weight
        per_it += 2 * (d + 1) * V * M * 4  # blur read + write
        per_it += N * (d + 1) * 8      # slice idx + weight
        per_it += N * M * 4            # write message
    per_it += N * M * 4                # write Q
    return T * per_it


def cu_masked_stream(device, lo, hi):
    """A HIP stream whose kernels run on compute units [lo, hi) only (A/B experiments; ctypes on the HIP runtime the
    library itself is linked against)."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipSetDevice(int(device))
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return st.value


class Workload:
    def __init__(self, device, batch, precision, workload, seed, arch="resnet50", share=None):
        import numpy as np

        from wsscam import _lib, synth

        global S, GFLOP_PER_IMAGE
        S = INPUT_SIZE_BY_ARCH[arch]
        GFLOP_PER_IMAGE = GFLOP_PER_IMAGE_BY_ARCH[arch]

        self.np, self._lib = np, _lib
        self.B, self.workload, self.device = batch, workload, device
        # A/B: WSC_BENCH_CUMASK="a[,b]" gives the conv stream compute units [0, a) and the lattice-build / mean-field streams
        # [b, 256) (b = a when omitted) through hipExtStreamCreateWithCUMask -- mask bits interleave over the 8 XCDs, so a
        # prefix is balanced.  Unset (default): ordinary streams, every kernel may use every CU.
        sA = sB = sC = None
        cm = os.environ.get("WSC_BENCH_CUMASK")
        if cm:
            a = int(cm.split(",")[0])
            b = int(cm.split(",")[1]) if "," in cm else a
            sA = cu_masked_stream(device, 0, a)
            sB = cu_masked_stream(device, b, 256)
            sC = cu_masked_stream(device, b, 256)
        self.ctx = _lib.Context(device, stream=sA)
        # second context (own stream) for the lattice build: it needs only the RGB images, so it runs
        # concurrently with the CNN forward pass of the same batch and joins before the inference
        self.ctx_build = _lib.Context(device, stream=sB)
        # third context: the mean-field loop.  With --pipeline (default) step i's loop runs here while
        # step i+1's conv stack (self.ctx) and lattice build (self.ctx_build) are already under way.
        self.ctx_crf = _lib.Context(device, stream=sC)
        # A/B: WSC_BENCH_OPT="7=0,..." sets path selectors (include/wsscam.h wsc_option) on the three contexts
        for kv in filter(None, os.environ.get("WSC_BENCH_OPT", "").split(",")):
            o, v = kv.split("=")
            for c in (self.ctx, self.ctx_build, self.ctx_crf):
                c.set_option(int(o), int(v))
        self.pending = None  # lattices of the step whose mean-field loop is still in flight
        # step_pipelined(): the loops of the last TWO steps may be in flight.  Marker p of the mean-field context is recorded
        # behind the loop that last used parity p's buffers (wsc_ctx_mark), so the host can wait for exactly that loop (two
        # steps old: normally finished) instead of the newest one -- the mean-field stream always has the next loop queued
        # behind the running one.  (Rounds 4-5 used two extra streams that only waited: with more streams than hardware queues
        # their wait packets could sit in front of the conv / build stream's kernels.  WSC_BENCH_DONE_STREAMS=1: that form.)
        self.done = [_lib.Context(device), _lib.Context(device)] if os.environ.get("WSC_BENCH_DONE_STREAMS") == "1" else None
        self.inflight = [None, None]
        prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[precision]
        # seeded random weights of the named architecture (no checkpoints offline), wsscam.synth
        if arch == "resnet50":
            self.sd = synth.resnet50_cam_state_dict(NUM_CLASSES, seed=0)
            self.net = _lib.Net(self.ctx, _lib.ARCH_RESNET50_CAM, self.sd, NUM_CLASSES, prec)
        elif arch == "vgg16":
            self.sd = synth.plain_state_dict("vgg16", NUM_CLASSES, True, seed=0)
            self.net = _lib.Net(self.ctx, _lib.ARCH_VGG16_CAM, self.sd, NUM_CLASSES, prec)
        else:
            self.sd = synth.plain_state_dict("m7", NUM_CLASSES, True, seed=0)
            rngw = np.random.default_rng(5)
            self.sd["gradcam_weights"] = rngw.normal(0, 0.05, (256, NUM_CLASSES)).astype(np.float32)
            self.net = _lib.Net(self.ctx, _lib.ARCH_M7_CAM, self.sd, NUM_CLASSES, prec)
        self.arch = arch
        self.h = self.net.cam_size(S)
        if share is not None:  # same images / labels as another workload (other precision)
            self.x_host, self.rgb_host, self.sizes, self.keys = share.x_host, share.rgb_host, share.sizes, share.keys
            self.native = share.native
        else:
            # real image-level labels (rows of the reference's voc12/cls_labels.npy in train_aug.txt order: the K
            # distribution of the dataset), synthetic VOC-like images
            labels = np.load(os.path.join(ROOT, "tests", "golden", "resnet50_cam.npz"))["trainaug_labels"]
            self.x_host, self.rgb_host, self.sizes, self.native = synth.image_batch(batch, S, seed, with_native=True)
            self.keys = [np.nonzero(labels[(i + seed * batch) % len(labels)])[0].astype(np.int32) for i in range(batch)]
        ctx = self.ctx
        self.x_dev = ctx.to_device(self.x_host)
        self.rgb_dev = ctx.to_device(self.rgb_host)
        self.x_devs, self.rgb_devs = [self.x_dev], [self.rgb_dev]  # resident batches; add_resident_batches() adds more
        self.cam_dev = ctx.alloc(batch * NUM_CLASSES * self.h * self.h * 4)
        # native-size make_cam outputs
        self.s_tot = sum(len(k) * ((H - 1) // 4 + 1) * ((W - 1) // 4 + 1) for k, (H, W) in zip(self.keys, self.sizes))
        self.h_tot = sum(len(k) * H * W for k, (H, W) in zip(self.keys, self.sizes))
        self.strided_dev = ctx.alloc(max(self.s_tot, 1) * 4)
        self.highres_dev = ctx.alloc(max(self.h_tot, 1) * 4)
        # CRF stack
        N = S * S
        # unaries / labels are double-buffered: step i+1 writes its unaries while step i's loop reads its own
        self.unary_bufs = [ctx.alloc(batch * ((NUM_CLASSES + 1 + 3) // 4 * 4) * N * 4) for _ in range(2)]
        self.label_bufs = [ctx.alloc(batch * N * 4) for _ in range(2)]
        self.unary_dev, self.label_dev = self.unary_bufs[0], self.label_bufs[0]
        self.parity = 0
        self.vg = self.vb = None
        self.e2e = None

    def add_resident_batches(self, n, seed):
        """n - 1 more batches of DIFFERENT images (same native sizes and labels, so the tail's geometry is unchanged) for the
        steady-state leg: the step then cannot live on inputs that stay in the 256 MB Infinity Cache."""
        from wsscam import synth

        for j in range(1, n):
            x, rgb, sizes = synth.image_batch(self.B, S, seed + j * len(synth.VOC_SIZES))
            assert sizes == self.sizes
            self.x_devs.append(self.ctx.to_device(x))
            self.rgb_devs.append(self.ctx.to_device(rgb))

    def next_batch(self):
        self.bi = (getattr(self, "bi", 0) + 1) % len(self.x_devs)
        self.x_dev, self.rgb_dev = self.x_devs[self.bi], self.rgb_devs[self.bi]

    def close(self):
        self.drain()
        for c in (self.ctx, self.ctx_build, self.ctx_crf) + (tuple(self.done) if self.done else ()):
            c.sync()
        self.net.close()

    # -- pieces ------------------------------------------------------------------------------
    def run_cnn(self):
        self.net.forward_cam(self.x_dev, self.B, S, self.cam_dev, None)

    def run_tail(self):
        self.tail_meta = self._lib.cam_postprocess(self.ctx, self.cam_dev, self.B, NUM_CLASSES, self.h, self.h, self.sizes,
                                                   self.keys, self.strided_dev, self.highres_dev)

    def run_unary(self):
        # upsample all 20 class maps to S x S, x /= max + 1e-5, [bg = 0.15 | maps] -> unaries: one fused call
        # (wsc_cam_postprocess + wsc_unary_from_maps give the same bits through 0.5 GB of intermediate maps)
        # written pixel-major ([B][N][24]): the layout the mean-field loop reads, no transpose pass in wsc_crf_inference
        self._lib.cam_unary(self.ctx, self.cam_dev, self.B, NUM_CLASSES, self.h, self.h, S, S, 0.15, self.unary_dev,
                            pixel_major=True)

    def crf_create(self):
        return self._lib.Crf(self.ctx_build, self.rgb_dev, self.B, S, S, CRF_CFG[0], CRF_CFG[2], CRF_CFG[3])

    def crf_infer(self, crf, ctx=None):
        crf.inference(self.unary_dev, NUM_CLASSES + 1, CRF_CFG[1], CRF_CFG[4], CRF_CFG[5], None, self.label_dev,
                      ctx=ctx or self.ctx, pixel_major=True)

    def drain(self):
        """Wait for the mean-field loop still in flight (pipelined mode) and release its lattices."""
        if self.pending is not None:
            self.ctx_crf.sync()
            self.pending.close()
            self.pending = None
        for p in (0, 1):
            self._retire(p)

    def _retire(self, p):
        """Wait for the loop that last used parity p's unary / label buffers and release its lattices."""
        if self.inflight[p] is not None:
            if self.done:
                self.done[p].sync()
            else:
                self.ctx_crf.wait_mark(p)
            self.inflight[p].close()
            self.inflight[p] = None

    def step_pipelined(self):
        """Same work per step as step(), issued so that consecutive steps overlap:
        stream A (ctx)       conv stack, tail, unaries of step i
        stream B (ctx_build) lattice build of step i
        stream C (ctx_crf)   mean-field loop of step i, after A and B  -- still running while the host
                             already enqueues step i+1 on A and B."""
        if self.pending is not None:
            self.drain()                 # (a loop left by another step function)
        self.parity ^= 1
        p = self.parity
        self._retire(p)                  # step i-2's loop read / wrote this parity's buffers: done before they are rewritten
        self.unary_dev, self.label_dev = self.unary_bufs[p], self.label_bufs[p]
        self.run_cnn()
        crf = self.crf_create()          # host blocks on the build stream only; A and C keep running
        if self.vg is None:
            self.vg, self.vb = crf.lattice_sizes()
        self.run_tail()
        self.run_unary()
        if os.environ.get("WSC_BENCH_INFLIGHT") == "1":
            self._retire(p ^ 1)          # (A/B: at most one loop in flight -- the host waits for step i-1's loop: round 3's schedule)
        self.ctx_crf.wait_for(self.ctx)
        self.ctx_crf.wait_for(self.ctx_build)
        self.crf_infer(crf, ctx=self.ctx_crf)   # queued behind step i-1's loop on the same stream: no host round trip between them
        if self.done:
            self.done[p].wait_for(self.ctx_crf)
        else:
            self.ctx_crf.mark(p)
        self.inflight[p] = crf

    def step(self, sequential=False):
        """One step, finished before it returns.  sequential=True additionally keeps the stages from
        overlapping each other (used for the per-kernel HIP-event measurement, where a kernel's duration
        must not include time shared with another stream's kernels)."""
        # stream 1: conv stack (enqueued asynchronously, returns at once)
        self.run_cnn()
        if self.workload != "cam_crf":
            self.run_tail()
            return
        if sequential:
            self.ctx.sync()
        # stream 2: lattice build of the same batch (needs only the RGB images) while the conv stack runs
        crf = self.crf_create()
        if self.vg is None:
            self.vg, self.vb = crf.lattice_sizes()
        # stream 1 again: tail + unaries, then join and run the mean-field loop
        self.run_tail()
        self.run_unary()
        self.ctx.wait_for(self.ctx_build)
        self.crf_infer(crf)
        self.ctx.sync()  # the lattice memory goes back to the build ctx's cache only when inference is done
        crf.close()

    # -- SURVEY config 3, second variant: M = K + 1 ---------------------------------------------------------
    def setup_kplus1(self):
        """Images grouped by K = number of positive classes; one CRF batch per group with M = K + 1 (background + the
        image's own classes: what cam_to_ir_label / eval_cam work with, 03b_irn/step/cam_to_ir_label.py:27-41)."""
        np = self.np
        order = sorted(range(self.B), key=lambda b: len(self.keys[b]))
        self.kp_rgb_dev = self.ctx.to_device(self.rgb_host[order])
        self.kp_x_dev = self.ctx.to_device(self.x_host[order])
        self.kp_keys = [self.keys[b] for b in order]
        self.kp_groups = []
        b0 = 0
        while b0 < self.B:
            K = len(self.kp_keys[b0])
            b1 = b0
            while b1 < self.B and len(self.kp_keys[b1]) == K:
                b1 += 1
            self.kp_groups.append((b0, b1 - b0, K))
            b0 = b1
        N = S * S
        self.kp_maps_dev = self.ctx.alloc(self.B * 6 * N * 4 + 4)
        self.kp_strided_dev = self.ctx.alloc(self.B * 6 * ((S - 1) // 4 + 1) ** 2 * 4 + 4)
        self.kp_unary_bufs = [self.ctx.alloc(self.B * 7 * N * 4) for _ in range(2)]  # double-buffered like the M = 21 step
        self.kp_label_bufs = [self.ctx.alloc(self.B * N * 4) for _ in range(2)]
        assert max(len(k) for k in self.kp_keys) <= 6 and min(len(k) for k in self.kp_keys) >= 1

    def step_kplus1(self):
        """One step of the M = K + 1 variant, pipelined like step_pipelined(): ONE ragged CRF object for the whole batch
        (wsc_crf_v: every image its own class count), its lattices built on stream B while the conv stack runs on A, its
        mean-field loop on C while the host already enqueues the next step."""
        N = S * S
        self.parity ^= 1
        p = self.parity
        self.net.forward_cam(self.kp_x_dev, self.B, S, self.cam_dev, None)
        crf = self._lib.CrfV(self.ctx_build, [self.kp_rgb_dev.ptr + b * N * 3 for b in range(self.B)], [(S, S)] * self.B,
                             CRF_CFG[0], CRF_CFG[2], CRF_CFG[3])
        # make_cam tail for every image's own classes at S x S (maps packed back to back), then [bg | maps] unaries per K-group
        self._lib.cam_postprocess(self.ctx, self.cam_dev, self.B, NUM_CLASSES, self.h, self.h, [(S, S)] * self.B, self.kp_keys,
                                  self.kp_strided_dev, self.kp_maps_dev)
        u_ptrs, m_off, u_off = [], 0, 0
        for (b0, nb, K) in self.kp_groups:
            self._lib.unary_from_maps(self.ctx, self.kp_maps_dev.ptr + m_off * N * 4, nb, K, N, 0.15,
                                      self.kp_unary_bufs[p].ptr + u_off * N * 4)
            u_ptrs += [self.kp_unary_bufs[p].ptr + (u_off + i * (K + 1)) * N * 4 for i in range(nb)]
            m_off += nb * K
            u_off += nb * (K + 1)
        self.drain()
        self.ctx_crf.wait_for(self.ctx)
        self.ctx_crf.wait_for(self.ctx_build)
        crf.inference(u_ptrs, [len(k) + 1 for k in self.kp_keys], CRF_CFG[1], CRF_CFG[4], CRF_CFG[5], None,
                      [self.kp_label_bufs[p].ptr + b * N * 4 for b in range(self.B)], ctx=self.ctx_crf)
        self.pending = crf

    # -- end to end: host batch in, files out ---------------------------------------------------------------
    def setup_e2e(self, out_dir, n_writers=8, u8=False):
        """End-to-end leg: a host batch in, .npy files out, around the pipelined step.

        Who issues what (round 6; profiles/r06_step_timeline_e2e.txt is the trace that led here):
          main thread      step i's kernels on the conv / build / mean-field streams; the copy-in of batch i+1, issued only
                           after the host has SEEN conv stack i-1 finish (the device buffer it fills is free)
          cam copier       sleeps until tail i is done, then copies cam / high_res out
          finisher         sleeps until loop i is done, copies the label maps out, hands the 64 files to the writer pool
        No copy is ever handed to the runtime before what it waits for has happened.  Rounds 4-5 queued every copy at once behind
        stream waits: the runtime turns a copy's dependency into a poll command on the DMA engine's own queue, so the label copy of
        step i (waiting ~20 ms for loop i) held back the copy-in of batch i+2 queued behind it on the same engine, the next conv stack
        waited for THAT, and every other step ran its loop with nothing beside it (13.5 ms per step against 12.1 resident).
        Outputs live in a ring of R = 3 steps (page-locked buffers, cam / unary / label device buffers, lattices): with 2, step i
        could not be enqueued before step i-2's files were on disk, which is ~2 ms after loop i-2 -- itself sharing the GPU with
        conv stack i-1 -- has finished."""
        from concurrent.futures import ThreadPoolExecutor

        np = self.np
        if self.e2e is not None:
            for k in ("pool", "finisher", "camcopier"):
                self.e2e[k].shutdown(wait=True)
        self.u8_offs = np.concatenate(([0], np.cumsum([im.size for im in self.native]))).astype(np.int64)
        R = max(2, min(3, int(os.environ.get("WSC_BENCH_E2E_RING", "3"))))
        while len(self.unary_bufs) < R:
            self.unary_bufs.append(self.ctx.alloc(self.unary_bufs[0].nbytes))
            self.label_bufs.append(self.ctx.alloc(self.label_bufs[0].nbytes))
        mk = lambda: self._lib.Context(self.device)
        self.e2e = {"dir": out_dir, "pool": ThreadPoolExecutor(n_writers), "u8": u8, "ring": R, "step": 0, "fed": False,
                    "finisher": ThreadPoolExecutor(1), "camcopier": ThreadPoolExecutor(1),
                    "pin_u8": [self.ctx.host_alloc(int(self.u8_offs[-1])) for _ in range(2)],
                    "pin_in": [self.ctx.host_alloc(self.x_host.nbytes) for _ in range(2)],
                    "pin_out": [self.ctx.host_alloc((max(self.s_tot, 1) + max(self.h_tot, 1)) * 4) for _ in range(R)],
                    "pin_lab": [self.ctx_crf.host_alloc(self.B * S * S * 4) for _ in range(R)],
                    "strided": [self.strided_dev] + [self.ctx.alloc(max(self.s_tot, 1) * 4) for _ in range(R - 1)],
                    "highres": [self.highres_dev] + [self.ctx.alloc(max(self.h_tot, 1) * 4) for _ in range(R - 1)],
                    "stage": [None, None], "fin": [None] * R, "crf": [None] * R,
                    "x": [self.ctx.alloc(self.x_host.nbytes) for _ in range(2)],
                    "u8d": [self.ctx.alloc(int(self.u8_offs[-1])) for _ in range(2)],
                    # copy streams: batch in / cam + high_res out / label maps out (never given a device-side wait)
                    "io": mk(), "oc": mk(), "ol": mk()}
        os.makedirs(out_dir, exist_ok=True)

    def _e2e_save(self, p, b, s_off, h_off, shapes):
        np = self.np
        e = self.e2e
        K, h4, w4, H0, W0 = shapes[b]
        st = e["pin_out"][p].view((max(self.s_tot, 1),), np.float32)
        hi = e["pin_out"][p].view((max(self.h_tot, 1),), np.float32, offset_bytes=max(self.s_tot, 1) * 4)
        from wsscam.step.make_cam import save_npy_array, save_npy_object  # np.load-compatible; one C call per file, no GIL

        save_npy_object(os.path.join(e["dir"], "img%03d.npy" % b),
                {"keys": self.keys[b].astype(np.int64), "cam": st[s_off[b]:s_off[b] + K * h4 * w4].reshape(K, h4, w4),
                 "high_res": hi[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0)})  # make_cam.py:80-82
        lab = e["pin_lab"][p].view((self.B, S, S), np.int32)[b]
        save_npy_array(os.path.join(e["dir"], "img%03d_crf.npy" % b), lab.astype(np.uint8))

    def _e2e_copy_cam(self, p):
        """Cam-copier thread: sleeps until the tail of the step in ring slot p has run (marker 2 + p of the conv stream), then
        copies cam / high_res of that slot into page-locked memory."""
        e = self.e2e
        self.ctx.wait_mark(2 + p)
        oc = e["oc"]
        oc.d2h_async(e["pin_out"][p], e["strided"][p], max(self.s_tot, 1) * 4)
        oc.d2h_async(e["pin_out"][p], e["highres"][p], max(self.h_tot, 1) * 4, dst_offset=max(self.s_tot, 1) * 4)
        oc.sync()

    def _e2e_finish(self, p, cam_copy, s_off, h_off, shapes):
        """Finisher thread: sleeps until the mean-field loop of the step in ring slot p is done (marker p of its stream), copies
        the label maps out, and hands the 64 files of the step to the writer pool; returns their futures."""
        e = self.e2e
        self.ctx_crf.wait_mark(p)
        ol = e["ol"]
        ol.d2h_async(e["pin_lab"][p], self.label_bufs[p], self.B * S * S * 4)
        ol.sync()
        cam_copy.result()
        return [e["pool"].submit(self._e2e_save, p, b, s_off, h_off, shapes) for b in range(self.B)]

    def _e2e_retire(self, p):
        """The step that last used ring slot p (R steps ago): its files are on disk, its lattices can go."""
        e = self.e2e
        if e["fin"][p] is not None:
            for f in e["fin"][p].result():
                f.result()
            e["fin"][p] = None
        if e["crf"][p] is not None:
            e["crf"][p].close()
            e["crf"][p] = None

    def step_e2e(self):
        """step_pipelined() with the host boundary of the reference around it: a pageable float32 batch (what the
        DataLoader hands over) is staged through page-locked memory and copied in; cam / high_res and the label maps
        are copied out and written as .npy files by writer threads while the next steps compute."""
        np = self.np
        e = self.e2e
        if self.pending is not None or self.inflight[0] is not None or self.inflight[1] is not None:
            self.drain()  # (loops left by another step function)
        R = e["ring"]
        p = e["step"] % R                # ring slot of the step's outputs
        q = e["step"] % 2                # parity of its input buffers
        e["step"] += 1
        self.parity = q
        tr = e.setdefault("trace", {}) if os.environ.get("WSC_BENCH_E2E_TRACE") else None
        t = [time.perf_counter()]
        self._e2e_retire(p)              # the files of step i-R are on disk: its staging buffers are free again
        t.append(time.perf_counter())
        self.unary_dev, self.label_dev = self.unary_bufs[p], self.label_bufs[p]
        strided_keep, highres_keep = self.strided_dev, self.highres_dev
        self.strided_dev, self.highres_dev = e["strided"][p], e["highres"][p]
        if not e["fed"]:
            self._e2e_feed(q)            # (first step: nobody has fed this one)
            e["fed"] = True
        # batch i is on the device: its copy-in was issued a step ago, the host just checks.  (No marker / device-side wait on a
        # copy stream: a marker is a barrier packet in the stream's hardware queue, the runtime maps all streams onto four of
        # those, and behind another stream's mean-field kernels the marker -- and the conv stack waiting for it -- sat ~1 ms)
        e["io"].sync()
        x_keep, self.x_dev = self.x_dev, e["x"][q]
        if e["u8"]:
            # decoded images in: the dataset transform (resize, normalise, flip pair: a 45 us kernel) is the first launch of the
            # step's conv stream (on the copy stream it had to find free compute units between the conv stack's workgroups, and
            # the u8 leg, which moves 6x fewer bytes, was the slower one: VERDICT r5 weak #9)
            self._lib.msf_input_u8(self.ctx, e["u8d"][q], [im.shape[:2] for im in self.native], self.u8_offs[:-1], S,
                                   (104.0, 117.0, 123.0), (255.0, 255.0, 255.0), e["x"][q], pre_div255=False, pair=True)
        self.run_cnn()
        self.ctx.mark(q)                 # conv stack i has read device input buffer q
        self.x_dev = x_keep
        crf = self.crf_create()
        self.run_tail()
        self.ctx.mark(2 + p)
        cam_copy = e["camcopier"].submit(self._e2e_copy_cam, p)
        _, _, s_off, h_off, shapes = self.tail_meta
        self.strided_dev, self.highres_dev = strided_keep, highres_keep
        self.run_unary()
        self.ctx_crf.wait_for(self.ctx)
        self.ctx_crf.wait_for(self.ctx_build)
        self.crf_infer(crf, ctx=self.ctx_crf)  # queued behind step i-1's loop, no host round trip
        self.ctx_crf.mark(p)
        e["crf"][p] = crf
        e["fin"][p] = e["finisher"].submit(self._e2e_finish, p, cam_copy, s_off, h_off, shapes)
        t.append(time.perf_counter())
        # batch i + 2 -> the page-locked staging buffers batch i came from (its copy-in, issued a step ago, has long finished);
        # batch i + 1 -> device input buffer q^1 once conv stack i-1, which read it, is done
        e["stage"][q] = self._e2e_stage(q)
        t.append(time.perf_counter())
        self.ctx.wait_mark(q ^ 1)
        t.append(time.perf_counter())
        self._e2e_feed(q ^ 1)
        t.append(time.perf_counter())
        if tr is not None:  # host time of the step by phase (profiles/e2e_probe.py prints the averages)
            for k, v in zip(("retire", "enqueue", "stage_submit", "wait_conv", "feed"), np.diff(t)):
                tr[k] = tr.get(k, 0.0) + float(v)
            tr["steps"] = tr.get("steps", 0) + 1

    def _e2e_feed(self, q):
        """Copy-in of the batch staged in parity q's page-locked buffers into device input buffer q (the caller has seen the
        conv stack that last read it finish)."""
        e = self.e2e
        for f in (e["stage"][q] or self._e2e_stage(q)):
            f.result()
        e["stage"][q] = None
        io = e["io"]
        if e["u8"]:
            io.h2d_async(e["u8d"][q], e["pin_u8"][q], int(self.u8_offs[-1]))
        else:
            io.h2d_async(e["x"][q], e["pin_in"][q], self.x_host.nbytes)

    def _e2e_stage(self, p):
        """The DataLoader's batch (pageable memory) -> parity p's page-locked staging buffers, on the pool."""
        np = self.np
        e = self.e2e
        if e["u8"]:
            buf = e["pin_u8"][p].view((int(self.u8_offs[-1]),), np.uint8)

            def _cp(k0, k1):
                for k in range(k0, k1):
                    buf[self.u8_offs[k]:self.u8_offs[k + 1]] = self.native[k].reshape(-1)

            nn = len(self.native)
            nchunk = min(8, nn)
            return [e["pool"].submit(_cp, nn * c // nchunk, nn * (c + 1) // nchunk) for c in range(nchunk)]
        # float32: the 79 MB copy is split over the pool (numpy releases the interpreter lock for the copy; one thread
        # moves ~9 GB/s)
        dst = e["pin_in"][p].view(self.x_host.shape, np.float32)
        nchunk = min(8, self.x_host.shape[0])
        bounds = [self.x_host.shape[0] * c // nchunk for c in range(nchunk + 1)]
        return [e["pool"].submit(np.copyto, dst[bounds[c]:bounds[c + 1]], self.x_host[bounds[c]:bounds[c + 1]])
                for c in range(nchunk)]

    def drain_e2e(self):
        e = self.e2e
        for k in range(e["ring"]):
            self._e2e_retire((e["step"] + k) % e["ring"])
        self.ctx.sync()
        for c in (e["io"], e["oc"], e["ol"]):
            c.sync()
        for st in e["stage"]:
            for f in st or []:
                f.result()
        e["stage"] = [None, None]
        e["fed"] = False

    def timed(self, fn, reps):
        """Average device time of fn() over reps, HIP events on the ctx stream."""
        fn()  # untimed: the first call on this ctx may grow its workspace arena (a hipMalloc of a few GB)
        self.ctx.sync()
        self.ctx.timer_begin()
        for _ in range(reps):
            fn()
        return self.ctx.timer_end() / reps


def cpu_baseline(wl, budget_s, product_labels=None):
    """-> (cpu_baseline object, parity object or None).
    Oracle on the host cores (the only place bench.py touches oracle/): torch-CPU fp32 restatement of
    make_cam._work (batch = 1 image, as the reference runs it) + the C dense-CRF restatement, once single-threaded
    (pydensecrf is single-threaded per image) and once over images in parallel (one image per thread: what an
    OpenMP-over-images loop around the reference's per-image CRF gives).  kind = "port": the reference itself cannot
    travel to the GPU box.  The label maps the oracle chain produces for the first images of the batch are compared with
    `product_labels` (the product chain's labels of the same images, int32 [B][S*S]): the `parity` object."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from oracle import cnn_ref
    from tests import helpers

    sd = {k: torch.from_numpy(v) for k, v in wl.sd.items()}  # the product run's own weights
    # pick the thread count the way a user of the reference would: the fastest of a few settings
    # (all cores of a many-core host is slower than 16-64 threads on these small batch-1 convolutions)
    ncpu = os.cpu_count() or 1
    best = None
    x0 = torch.from_numpy(wl.x_host[0])
    for nt in sorted({min(ncpu, v) for v in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            cnn_ref.resnet50_cam_forward(x0, sd)  # warm-up / primitive creation
            t0 = time.perf_counter()
            cnn_ref.resnet50_cam_forward(x0, sd)
            dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    n_cam, t_cam = 0, 0.0
    cams = []
    while n_cam < wl.B and (n_cam < 2 or t_cam < budget_s * 0.4):
        x = torch.from_numpy(wl.x_host[n_cam])
        lab = torch.zeros(NUM_CLASSES)
        lab[torch.from_numpy(wl.keys[n_cam].astype(np.int64))] = 1
        t0 = time.perf_counter()
        with torch.no_grad():
            cam = cnn_ref.resnet50_cam_forward(x, sd)
            valid = torch.nonzero(lab)[:, 0]
            cnn_ref.make_cam_tail(cam, wl.sizes[n_cam], valid)
        t_cam += time.perf_counter() - t0
        cams.append(cam)
        n_cam += 1
    if wl.workload != "cam_crf":
        return {"value": round(n_cam / t_cam, 4), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": "%d images torch-CPU fp32 ResNet50-CAM+tail (%.2f s/img, %d threads)" % (n_cam, t_cam / n_cam, cores)}, None

    def unary_of(i):
        with torch.no_grad():
            _, hi = cnn_ref.make_cam_tail(cams[i % n_cam], (S, S), torch.arange(NUM_CLASSES))
        v = np.concatenate([np.full((1, S * S), 0.15, np.float32), hi.numpy().reshape(NUM_CLASSES, -1)], 0)
        return -np.log(np.clip(v / v.sum(0, keepdims=True), 1e-5, 1.0)).astype(np.float32)

    t0 = time.perf_counter()
    helpers.crf_oracle(wl.rgb_host[0], unary_of(0), CRF_CFG)
    t_crf1 = time.perf_counter() - t0                                   # one image, one thread
    threads = max(1, min(ncpu, 64))
    n_par = max(threads, int(budget_s * 0.4 / max(t_crf1, 1e-3)) * threads)
    n_par = min(n_par, 4 * threads)
    n_lab = min(n_cam, 8)                                               # images whose oracle labels are kept (image i with ITS unaries)
    us = [unary_of(i) for i in range(n_lab)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL inside the C oracle
        res = list(ex.map(lambda i: helpers.crf_oracle(wl.rgb_host[i % n_lab], us[i % n_lab], CRF_CFG)[1], range(n_par)))
    crf_rate = n_par / (time.perf_counter() - t0)                       # images/s, `threads` images in flight
    per_img = t_cam / n_cam + 1.0 / crf_rate
    parity = None
    if product_labels is not None:
        n_chk = min(n_lab, n_par)
        parity = helpers.label_parity(product_labels[:n_chk], np.stack(res[:n_chk]), NUM_CLASSES + 1)
        parity = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in parity.items()}
        parity["against"] = ("all-fp32 oracle chain (torch-CPU ResNet50-CAM -> make_cam tail -> [bg | maps] unaries -> C dense-CRF) "
                             "on the first %d images of this batch; labels scored as eval_cam scores pseudo-labels" % n_chk)
    return {"value": round(1.0 / per_img, 4), "unit": "images/s", "cores": max(cores, threads), "kind": "port",
            "value_crf_single_thread": round(1.0 / (t_cam / n_cam + t_crf1), 4),
            "sample": "%d images torch-CPU fp32 ResNet50-CAM+tail (%.3f s/img, %d threads) + C dense-CRF M=21 T=10: %d images over "
                      "%d threads, one image per thread (%.1f images/s; %.2f s/img on one thread)"
                      % (n_cam, t_cam / n_cam, cores, n_par, threads, crf_rate, t_crf1)}, parity


def build_roofline(prof, n_steps, precision, traffic_classes=None, loop_gbps=None):
    """-> (kernels, roofline) from a per-class HIP-event profile {class: (launches, total ms, algorithmic work)} of `n_steps`
    steps whose stages did not overlap.  `roofline` describes the class with the largest time per step and carries the two
    fractions the north star names as sub-objects:
      conv_mfma -- the whole conv stack (every conv_igemm launch of a step): algorithmic FLOPs (one multiply-add per term)
                   over the summed kernel time against the peak of the precision mode; `executed` counts the three MFMA
                   products per term of the split modes against the dense 16-bit peak
      crf_hbm   -- update_splat_kernel, the mean-field loop's dominant kernel: the SURVEY 8(d) bytes it covers per launch
                   against 8 TB/s; `traffic` / `frac_traffic`: HBM bytes per launch from the committed PMC passes
                   (profiles/hbm_traffic.json -- counters cannot be read from inside the process)."""
    traffic_classes = traffic_classes or {}
    kernels = {}
    for name, (calls, ms, work) in prof.items():
        is_conv = name.startswith("conv_igemm")
        rate = work / (ms * 1e-3) / (1e12 if is_conv else 1e9) if ms > 0 else 0.0
        kernels[name] = {"launches_per_step": calls // n_steps, "avg_us": round(ms / calls * 1e3, 2),
                         "ms_per_step": round(ms / n_steps, 4), ("TFLOP/s" if is_conv else "GB/s"): round(rate, 2)}
    # the dominant KERNEL: conv_igemm_kernel is one __global__ template whose instantiations (tile shapes) the profiler lists
    # as separate classes -- they count together, like the three instantiations of update_splat_kernel do in their class
    groups = {}
    for n, (c, m, w) in prof.items():
        if n == "crf_build(all)":
            continue
        g = "conv_igemm_kernel" if n.startswith("conv_igemm") else n
        gc, gm, gw = groups.get(g, (0, 0.0, 0.0))
        groups[g] = (gc + c, gm + m, gw + w)
    dom = max(groups, key=lambda n: groups[n][1])
    calls, ms, work = groups[dom]
    if dom.startswith("conv_igemm"):
        roofline = {"bound": "mfma", "achieved": round(work / (ms * 1e-3) / 1e12, 2), "peak": round(PEAK_TFLOPS[precision], 1),
                    "unit": "TFLOP/s", "traffic": None}
    else:
        roofline = {"bound": "hbm", "achieved": round(work / (ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "traffic": None}
    roofline["frac"] = round(roofline["achieved"] / roofline["peak"], 4)
    roofline["kernel"] = dom
    roofline["algorithmic_bytes_per_launch" if roofline["bound"] == "hbm" else "flop_per_launch"] = round(work / calls)

    def add_traffic(obj, cls_name, per_launch_us):
        cls = traffic_classes.get(cls_name)
        if not cls:
            return
        obj["traffic"] = cls["bytes_per_launch"]
        obj["frac_traffic"] = round(cls["bytes_per_launch"] / (per_launch_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
        obj["traffic_source"] = ("committed: profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this "
                                 "command, %s); not re-measured in this run" % cls.get("round", "r02"))

    roofline["avg_launch_us"] = round(ms / calls * 1e3, 2)
    roofline["launches_per_step"] = calls // n_steps
    add_traffic(roofline, dom, ms / calls * 1e3)  # (conv: HBM bytes per launch averaged over every instantiation's launches)
    for name, k in kernels.items():               # every class the PMC passes cover: measured HBM bytes per launch beside its time
        cls = traffic_classes.get(name)
        if cls:
            k["traffic_bytes_per_launch"] = cls["bytes_per_launch"]
            k["traffic_GBps"] = round(cls["bytes_per_launch"] / (k["avg_us"] * 1e-6) / 1e9, 1) if k["avg_us"] > 0 else None
    conv = [(c, m, w) for n, (c, m, w) in prof.items() if n.startswith("conv_igemm")]
    if conv:
        c_ms, c_fl = sum(m for _, m, _ in conv), sum(w for _, _, w in conv)
        mult = 3 if precision in ("f16x3", "bf16x3") else 1
        roofline["conv_mfma"] = {"kernel": "conv_igemm_kernel + stem_pool_kernel (all %d launches of the conv stack)" % (sum(c for c, _, _ in conv) // n_steps),
                                 "achieved": round(c_fl / (c_ms * 1e-3) / 1e12, 2), "peak": round(PEAK_TFLOPS[precision], 1),
                                 "unit": "TFLOP/s", "frac": round(c_fl / (c_ms * 1e-3) / 1e12 / PEAK_TFLOPS[precision], 4),
                                 "ms_per_step": round(c_ms / n_steps, 4), "mfma_products_per_term": mult,
                                 "executed": round(mult * c_fl / (c_ms * 1e-3) / 1e12, 2), "peak_executed": 2500.0,
                                 "peak_note": "dense f16 MFMA peak 2500 TFLOP/s / %d products per algorithmic multiply-add" % mult}
    if "update_splat_kernel" in prof:
        uc, ums, uw = prof["update_splat_kernel"]
        roofline["crf_hbm"] = {"kernel": "update_splat_kernel", "achieved": round(uw / (ums * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(uw / (ums * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                               "algorithmic_bytes_per_launch": round(uw / uc), "avg_launch_us": round(ums / uc * 1e3, 2),
                               "launches_per_step": uc // n_steps, "traffic": None}
        add_traffic(roofline["crf_hbm"], "update_splat_kernel", ums / uc * 1e3)
        if loop_gbps is not None:
            roofline["crf_hbm"]["loop_algorithmic_GBps"] = loop_gbps
    return kernels, roofline


def hsn_measure(args, device):
    """BASELINE config 5 on one GPU: `segment_adp` (03c_hsn/demo.py:271-380) on batches of ADP-like patches, device
    resident from the batch upload to the label maps.  One step = one batch of --batch images (reference: 16)."""
    import numpy as np

    from wsscam import _lib, synth
    from wsscam.hsn import demo as hsn_demo
    from wsscam.net import vgg16_cam
    from wsscam.net.common import grad_cam_alpha

    C, S_ = 31, 321
    prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[args.precision]
    sd = synth.plain_state_dict("vgg16", C, batchnorm=False, seed=0)  # ADP models have no BatchNorm (vgg16_cam.py:16-19)
    rng = np.random.default_rng(4242)
    images = [synth.adp_image(rng, S_, S_) for _ in range(args.batch)]
    # classifier head calibrated to the random features (synth.specialise_classifier: with i.i.d. weights the Grad-CAM arg-max
    # degenerates to 2-4 morphological classes and 1 functional class per image): final feature maps of four patches
    model = vgg16_cam.CAM(None, "adp_morph", "ADP_VGG16", C, None, precision=prec)
    model.load_state_dict(sd)
    model.cuda(device)
    n_cal = min(4, args.batch)
    xs = np.stack([np.transpose((im.astype(np.float32) - 193.09203) / 56.450138, (2, 0, 1)) for im in images[:n_cal]])
    hf, F = model._net.cam_size(S_), model._net.feat_channels()
    fd = model.ctx.alloc(n_cal * hf * hf * F * 4)
    model._net.forward_features(model.ctx.to_device(np.ascontiguousarray(xs)), n_cal, S_, fd)
    synth.specialise_classifier(sd, "vgg16", C, model.ctx.to_host(fd, (n_cal, hf, hf, F), np.float32), seed=0)
    model = vgg16_cam.CAM(None, "adp_morph", "ADP_VGG16", C, None, precision=prec)
    model.load_state_dict(sd)
    model.cuda(device)
    alpha = grad_cam_alpha(sd["vgg16.classifier.0.weight"], S_ // 8, S_ // 8, "avg")
    thr = np.full((1, C), 0.5)
    cfgs = {"morph": np.array([3 / 2, 3, 80 / 2, 13, 10, 10]), "func": np.array([3 / 2, 3, 80 / 2, 13, 10, 10])}
    eff_m = {}

    def step(stats=None):
        return hsn_demo.segment_adp(model, alpha, thr, images, cfgs, S_, args.batch, stats=stats)

    for _ in range(max(args.warmup, 1)):
        eff_m = {}
        out = step(eff_m)
    # The timed region is ONE driver call over steps x batch patches (the reference's dataset loop, 03c_hsn/demo.py:318-380):
    # the driver keeps two batches in flight on two streams, so a batch's host decisions hide behind the other's kernels
    hsn_lanes = int(os.environ.get("WSC_BENCH_HSN_LANES", "3"))  # (segment_adp's default)
    chain = os.environ.get("WSC_BENCH_CHAIN", "0") == "1"  # (A/B: 1 = the lanes' conv stacks take turns; the driver's default is 0)
    hsn_demo.segment_adp(model, alpha, thr, images * 3, cfgs, S_, args.batch, n_lanes=hsn_lanes, chain_stacks=chain)  # (untimed: the other lanes' contexts and workspaces)
    model.ctx.sync()
    ctx = model.ctx
    t0 = time.perf_counter()
    hsn_demo.segment_adp(model, alpha, thr, images * args.steps, cfgs, S_, args.batch, n_lanes=hsn_lanes, chain_stacks=chain)
    ctx.sync()
    elapsed = time.perf_counter() - t0
    ctx.profile_begin()
    step()
    prof = ctx.profile_end()
    kernels, roofline = build_roofline(prof, 1, args.precision)
    m_classes = {h: sorted({int(len(np.unique(lab))) for lab in out[h]}) for h in out}
    extra = {}
    if not args.no_cpu_baseline:
        # the oracle chain (the only place this workload touches oracle/): torch-CPU VGG16 + numpy HSN post-processing + C
        # dense-CRF for both HTT types on a bounded sample; its label maps score the product's (parity)
        import torch

        from tests import helpers

        n_cpu = min(2, args.batch)
        sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        t0 = time.perf_counter()
        ref = helpers.oracle_chain_hsn_adp(images[:n_cpu], sdt, alpha, 0.5, cfgs)
        t_cpu = time.perf_counter() - t0
        extra["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                 "sample": "%d patches: torch-CPU fp32 VGG16 + numpy/scipy HSN post-processing + C dense-CRF x2 "
                                           "(morph + func, one thread), %.2f s/img" % (n_cpu, t_cpu / n_cpu)}
        par = {}
        for h in ("morph", "func"):
            n_cls = len(hsn_demo.ADPClasses().classes["valid_" + h])
            pp = helpers.label_parity([np.asarray(m) for m in out[h][:n_cpu]], [np.asarray(m) for m in ref[h]], n_cls)
            par[h] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in pp.items()}
        par["against"] = "all-fp32 oracle chain (tests/helpers.py::oracle_chain_hsn_adp) on the first %d patches" % n_cpu
        extra["parity"] = par
    return {
        "metric": "images/sec HistoSegNet CAM+CRF (BASELINE config 5, ADP-like 321x321 patches, morph + func label maps)",
        "value": round(args.batch * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "vgg16 (31 classes, no BN) HSN Grad-CAM + modify_by_htt + cs-gradcam + dense-CRF x2 (morph 29 / "
                               "func 5 classes, 10 iters), 321x321, batch %d; one segment_adp call over steps x batch patches, three batches in "
                               "flight (three streams)" % args.batch, "batch_images": args.batch,
                   "distinct_labels_per_image": m_classes,
                   "effective_M": {h: {"min": int(min(v)), "mean": round(float(np.mean(v)), 2), "max": int(max(v))} for h, v in eff_m.items()},
                   "effective_M_note": "classes with mass per image = the M its dense CRF runs with (dcrf_process keeps the classes "
                                       "whose class-specific Grad-CAM is not all zero, 03c_hsn/utilities.py:425); classifier head "
                                       "calibrated to the random features (wsscam.synth.specialise_classifier)"},
        "roofline": roofline, "stages": {"kernels": kernels}, **extra}


def run_hsn(args, device):
    print(json.dumps(hsn_measure(args, device)))


def irn_measure(device, precision, arch="resnet50", n_images=32, reps=3):
    """BASELINE config 4 (IRNet inference) at VOC size on one GPU: the make_sem_seg_labels driver (03b_irn/step/
    make_sem_seg_labels.py:22-143) on `n_images` 375 x 500 images -- EdgeDisplacement on the [orig, flip] pair zero-padded to
    512 x 512 (:46), boundary maps to the CAM size, the random walk of K = 2 strided CAMs at 94 x 125 (beta 10, 2^8 steps), x4
    upsample, / max, background channel, arg-max: host arrays in, label maps out, no file I/O.  Seeded random weights
    (wsscam.synth.irn_state_dict: no trained IRNet weights offline)."""
    import types

    import numpy as np

    from wsscam import _lib, synth
    from wsscam.step import make_sem_seg_labels as mssl

    prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[precision]
    sd = synth.irn_state_dict(arch, seed=0)
    if arch == "vgg16":
        from wsscam.net import vgg16_irn as irn_mod

        model = irn_mod.EdgeDisplacement(None, "voc12", "", 20, None, precision=prec)
    else:
        from wsscam.net import resnet50_irn as irn_mod

        model = irn_mod.EdgeDisplacement(None, 20, precision=prec)
    model.load_state_dict(sd, strict=False)
    model.cuda(device)
    rng = np.random.default_rng(0)
    h, w, K = 94, 125, 2
    n_images = int(os.environ.get("WSC_BENCH_IRN_BATCH", n_images))
    packs = [{"name": "i%d" % i, "img": rng.normal(0, 1, (2, 3, 375, 500)).astype(np.float32), "size": (375, 500)} for i in range(n_images)]
    cam_dicts = [{"keys": np.array([3, 11]), "cam": rng.random((K, h, w)).astype(np.float32)} for _ in range(n_images)]
    dargs = types.SimpleNamespace(dataset="voc12", beta=10, exp_times=8, sem_seg_bg_thres=0.25)
    # the driver's dataset loop (make_sem_seg_labels._work) over `reps` batches: two batches in flight on two streams
    n_lanes = int(os.environ.get("WSC_BENCH_IRN_LANES", "3"))  # (the driver's default, make_sem_seg_labels._work)
    chain = os.environ.get("WSC_BENCH_CHAIN", "0") == "1"  # (A/B: 1 = the lanes' network passes take turns; the driver's default is 0)
    mssl.sem_seg_batches(model, [(packs, cam_dicts)] * n_lanes, dargs, n_lanes=n_lanes, chain_stacks=chain)
    model.ctx.sync()
    t0 = time.perf_counter()
    mssl.sem_seg_batches(model, [(packs, cam_dicts)] * reps, dargs, n_lanes=n_lanes, chain_stacks=chain)
    model.ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(n_images / dt, 2), "unit": "images/s", "ms_per_image": round(dt / n_images * 1e3, 3), "dtype": precision,
            "workload": "IRNet inference (BASELINE config 4): %s EdgeDisplacement @512 pad -> random walk K=%d at %dx%d, 2^8 steps -> "
                        "label map at 375x500; make_sem_seg_labels driver loop, %d images per batch, three batches in flight, host arrays in / label maps out"
                        % (arch, K, h, w, n_images)}


def make_cam_measure(device, precision, n_small=288, n_large=1824, batch=32):
    """BASELINE config 1 (ResNet50 CAM over VOC2012 images at 321 x 321, batch 32) through the PRODUCT driver
    `step.make_cam.run(args)` (03b_irn/step/make_cam.py:95-124): a dataset of decoded 375 x 500 images (two positive classes each)
    -> loader threads -> page-locked lane -> resize / normalise / flip pair + conv stack + CAM head + native-size tail on the
    device -> `cam` / `high_res` copied out -> one .npy per image on a tmpfs.  Reported: the MARGINAL rate between a run over
    n_small and a run over n_large images (the constructor, weight packing and the first batches drop out).  Seeded random
    weights and synthetic images."""
    import shutil
    import tempfile

    import numpy as np

    from wsscam import _lib, synth
    from wsscam.step import make_cam

    prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f16x3": _lib.PREC_F16X3}[precision]
    sd = synth.resnet50_cam_state_dict(20, seed=0)
    rng = np.random.default_rng(0)
    imgs = [synth.synth_image(rng, 375, 500) for _ in range(batch)]
    labels = []
    for i in range(batch):
        lab = np.zeros(20, np.float32)
        lab[[i % 20, (7 * i + 3) % 20]] = 1
        labels.append(lab)
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") else None

    def once(n):
        packs = [{"name": "im%05d" % i, "img_u8": imgs[i % batch], "size": (375, 500), "label": labels[i % batch]} for i in range(n)]
        d = tempfile.mkdtemp(prefix="wsc_mc_", dir=tmp_root)
        a = argparse.Namespace(cam_network="net.resnet50_cam", model_dir=None, dataset="voc12", tag="", num_classes=20,
                               use_cls=list(range(20)), model_id="resnet50", state_dict=sd, split="train_aug", dataset_obj=packs,
                               cam_out_dir=d, outsize=(S, S), n_gpus=1, cam_device_ids=[device], cam_batch_images=batch,
                               cam_precision=prec, cam_weights_name="unused", norm_mode="int", val_list=None, dev_root=None,
                               cam_scales=(1.0,), class_names={"bg": ["background"], "fg": ["c%d" % i for i in range(20)]},
                               cam_pipeline_chain=os.environ.get("WSC_BENCH_CHAIN", "1") != "0")  # (A/B: 0 = interleaved stacks)
        try:
            t0 = time.perf_counter()
            make_cam.run(a)
            dt = time.perf_counter() - t0
            n_files = len(os.listdir(d))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if n_files != n:
            raise RuntimeError("make_cam wrote %d files for %d images" % (n_files, n))
        return dt

    once(2 * batch)  # (library, lanes, thread pools, page cache)
    if os.environ.get("WSC_BENCH_MAKE_CAM_ONE_RUN") == "1":  # (profiles/r06_make_cam_prof.sh: one long run under the kernel trace)
        t2 = once(n_large)
        return {"value": round(n_large / t2, 2), "unit": "images/s", "dtype": precision, "seconds": [round(t2, 3)], "images": [n_large],
                "workload": "make_cam driver, ONE run of %d images including the model set-up (trace mode)" % n_large}
    t1, t2 = min(once(n_small), once(n_small)), min(once(n_large), once(n_large))  # (best of two: the difference of two
    # sub-second runs moves by several per cent with a single late batch)
    rate = (n_large - n_small) / max(t2 - t1, 1e-9)
    return {"value": round(rate, 2), "unit": "images/s", "dtype": precision, "seconds": [round(t1, 3), round(t2, 3)], "images": [n_small, n_large],
            "workload": "make_cam driver (BASELINE config 1): ResNet50 CAM, decoded 375x500 uint8 images in, device transform to 321x321 + "
                        "flip pair, batch %d, three lanes; keys / cam (2,94,125) / high_res (2,375,500) per image written as .npy on %s; "
                        "marginal rate between the two runs" % (batch, "tmpfs" if tmp_root else "the system temp dir")}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not launched through torch.distributed.run: start it as a child (nothing touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch

    dist = None
    device = int(os.environ.get("WSC_BENCH_DEVICE", local_rank))  # override only for single-GPU dry runs
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")

    if args.workload == "make_cam":  # BASELINE config 1 through step.make_cam.run alone (also a leg of the default line)
        if rank == 0:
            print(json.dumps(make_cam_measure(device, args.precision)))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "hsn":
        if rank == 0:
            run_hsn(args, device)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    wl = Workload(device, args.batch, args.precision, args.workload, seed=rank, arch=args.arch)

    def barrier(w):
        w.ctx.sync()
        w.ctx_build.sync()
        w.ctx_crf.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    pipelined = args.workload == "cam_crf" and not args.no_pipeline

    def timed_run(w, do_step, steps, warmup, drain):
        """W warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides -> seconds."""
        for _ in range(warmup):
            do_step()
        drain()
        barrier(w)
        t0 = time.perf_counter()
        for _ in range(steps):
            do_step()
        drain()
        w.ctx.sync()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    do_step = wl.step_pipelined if pipelined else wl.step
    steps = args.steps
    if args.seconds > 0:  # steady state: size the timed region from a short probe (same K on every rank)
        probe = timed_run(wl, do_step, 3, max(args.warmup, 1), wl.drain) / 3
        steps = max(steps, int(args.seconds / probe) + 1)
        if dist is not None:
            t = torch.tensor([steps], dtype=torch.int64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steps = int(t.item())
    # strong scaling: the K steps are one fixed image set, sharded -- rank g owns ceil(K / N) of them
    my_steps = (steps + world - 1) // world if args.scaling == "strong" else steps
    elapsed = timed_run(wl, do_step, my_steps, args.warmup, wl.drain)
    # what the line says about the ranks behind it: who contributed, each rank's own time (the value is computed from the MAX)
    ranks_info = {"ranks_seen": 1, "barrier_world_size": 1, "backend": None, "launch": "single process",
                  "per_rank_ms_per_step": [round(elapsed / my_steps * 1e3, 4)],
                  "per_rank_value": [round(args.batch * my_steps / elapsed, 3)], "devices": [device]}
    if dist is not None:
        dev_t = "cuda" if args.dist_backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed, float(device), float(my_steps), float(rank)], dtype=torch.float64, device=dev_t)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = [[float(v) for v in e.cpu()] for e in every]
        assert sorted(int(e[3]) for e in every) == list(range(world)), "a rank is missing from the gather"
        ranks_info = {"ranks_seen": len(every), "barrier_world_size": dist.get_world_size(),
                      "backend": "%s (%s)" % (args.dist_backend, "RCCL over xGMI: timing barrier + max only, no data-path collective"
                                              if args.dist_backend == "nccl" else "CPU dry run"),
                      "launch": "torch.distributed.run, one process per GPU",
                      "per_rank_ms_per_step": [round(e[0] / e[2] * 1e3, 4) for e in every],
                      "per_rank_value": [round(args.batch * e[2] / e[0], 3) for e in every],
                      "devices": [int(e[1]) for e in every]}
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert abs(elapsed - max(e[0] for e in every)) < 1e-9
        dist.barrier()

    # ---- steady state: >= --steady-seconds of steps cycling over four distinct resident batches (rank 0 of a 1-GPU run) ----
    steady = None
    if world == 1 and args.workload == "cam_crf" and not args.quick and args.steady_seconds > 0:
        wl.add_resident_batches(4, seed=rank)

        def cycle_step():
            wl.next_batch()
            do_step()

        probe = timed_run(wl, cycle_step, 4, 2, wl.drain) / 4
        n_st = int(1.03 * args.steady_seconds / probe) + 1
        t_st = timed_run(wl, cycle_step, n_st, 0, wl.drain)
        steady = {"value": round(args.batch * n_st / t_st, 3), "steps": n_st, "seconds": round(t_st, 3),
                  "ms_per_step": round(t_st / n_st * 1e3, 4), "resident_batches": len(wl.x_devs)}
        wl.bi = len(wl.x_devs) - 1
        wl.next_batch()  # back to batch 0 for the stage measurements below

    # ---- per-stage device times (HIP events on the ctx stream), same resident inputs -------------
    reps = 5
    t_cnn = wl.timed(wl.run_cnn, reps)
    t_tail = wl.timed(wl.run_tail, reps)
    stages = {"cnn_ms": round(t_cnn, 4), "tail_ms": round(t_tail, 4),
              "conv_stack_tflops": round(GFLOP_PER_IMAGE * wl.B / t_cnn, 2)}
    if args.workload == "cam_crf":
        t_un = wl.timed(wl.run_unary, reps)
        t0c = time.perf_counter()
        crf = wl.crf_create()
        wl.ctx_build.sync()
        t_create = (time.perf_counter() - t0c) * 1e3
        wl.ctx.wait_for(wl.ctx_build)
        t_inf = wl.timed(lambda: wl.crf_infer(crf), 3)
        crf.close()
        vg, vb = float(wl.vg.mean()), float(wl.vb.mean())
        by = crf_bytes_per_image(S * S, NUM_CLASSES + 1, CRF_CFG[5], vg, vb)
        stages["sum_ms"] = round(t_cnn + t_tail + t_un + t_create + t_inf, 4)  # the step without any overlap
        stages.update({"unary_ms": round(t_un, 4), "crf_create_ms": round(t_create, 4), "crf_infer_ms": round(t_inf, 4),
                       "lattice_vertices_gauss": round(vg, 1), "lattice_vertices_bilat": round(vb, 1),
                       "crf_loop_algorithmic_GBps": round(by * wl.B / (t_inf * 1e-3) / 1e9, 2)})

    # ---- per-kernel roofline: every launch of two more steps bracketed by HIP events on its stream; the
    # stages of these steps do not overlap, so a duration is the kernel's own ----
    wl.ctx.profile_begin()
    wl.ctx_build.profile_begin()
    for _ in range(2):
        wl.step(sequential=True)
    prof = wl.ctx.profile_end()
    prof.update(wl.ctx_build.profile_end())
    # HBM bytes per launch of a class: NOT observed in this run -- FETCH_SIZE / WRITE_SIZE cannot be read from inside the
    # process; the figures are the ones committed under profiles/ from the PMC passes of the same command
    # (profiles/collect.sh -> profiles/hbm_traffic.json).  Only the default workload is profiled there.
    tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    default_wl = (args.workload == "cam_crf" and args.arch == "resnet50" and args.batch == 32 and args.precision == "f16x3")
    traffic_classes = {}
    if default_wl and os.path.exists(tj):
        with open(tj) as fh:
            traffic_classes = json.load(fh).get("classes", {})
    kernels, roofline = build_roofline(prof, 2, args.precision, traffic_classes,
                                       stages.get("crf_loop_algorithmic_GBps") if args.workload == "cam_crf" else None)
    stages["kernels"] = kernels

    # ---- the other numbers SURVEY 8(d) asks for, same images ----------------------------------------------------------------
    if args.workload == "cam_crf" and args.arch == "resnet50" and not args.quick:
        k_extra = max(3, min(args.steps, 10))
        if world == 1:
            if args.precision != "f16":  # the fast mode: half operands, one MFMA product per term (1.5e-2 on the CAM maps)
                w3 = Workload(device, args.batch, "f16", args.workload, seed=rank, arch=args.arch, share=wl)
                t3 = timed_run(w3, w3.step_pipelined if pipelined else w3.step, k_extra, 2, w3.drain)
                stages["value_f16"] = round(args.batch * k_extra / t3, 3)
                stages["cnn_ms_f16"] = round(w3.timed(w3.run_cnn, 3), 4)
                w3.close()
                del w3
            wl.setup_kplus1()
            tk = timed_run(wl, wl.step_kplus1, k_extra, 2, wl.drain)
            stages["value_M_eq_Kplus1"] = round(args.batch * k_extra / tk, 3)
            stages["M_eq_Kplus1_groups"] = ["K=%d x %d images" % (K, nb) for (_, nb, K) in wl.kp_groups]
        # End to end (host batch in, .npy files out) on EVERY rank at once: N workers share the host's cores, PCIe root and
        # page cache, so the aggregate is what a make_cam run over N GPUs sees.  Writers go to a tmpfs when
        # there is one (WSC_BENCH_TMP overrides): the measurement is the host pipeline, not a disk.
        import shutil
        import tempfile

        def agg(seconds):
            if dist is None:
                return seconds
            t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        tmp_root = os.environ.get("WSC_BENCH_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
        tmp = tempfile.mkdtemp(prefix="wsc_bench_r%d_" % rank, dir=tmp_root)
        try:
            from wsscam.step.pipeline import host_thread_budget

            n_wr = 8 if world == 1 else max(2, host_thread_budget(world)["n_writers"])
            # (30 steps: the timed region ends when the LAST step's files are on disk -- a tail of about one step that a
            # 10-step region charges at 10 %)
            k_e2e = max(k_extra, 30) if not args.quick else k_extra
            wl.setup_e2e(tmp, n_writers=n_wr)
            te = agg(timed_run(wl, wl.step_e2e, k_e2e, 3, wl.drain_e2e))
            stages["value_end_to_end"] = round(world * args.batch * k_e2e / te, 3)
            stages["end_to_end"] = ("per step and rank: 79 MB pageable float32 batch -> pinned -> H2D; D2H of cam + high_res (%.1f MB) and "
                                    "label maps (%.1f MB); %d .npy files through %d writer threads (%s); overlapped with the next step, whose staging copy the same threads make ahead of time; %d rank(s) at once, max over ranks"
                                    % ((wl.s_tot + wl.h_tot) * 4 / 1e6, args.batch * S * S * 4 / 1e6, 2 * args.batch, n_wr,
                                       "tmpfs" if tmp_root == "/dev/shm" else (tmp_root or "system temp dir"), world))
            wl.setup_e2e(tmp, n_writers=n_wr, u8=True)
            te = agg(timed_run(wl, wl.step_e2e, k_e2e, 3, wl.drain_e2e))
            stages["value_end_to_end_u8_input"] = round(world * args.batch * k_e2e / te, 3)
            stages["end_to_end_u8_input"] = ("as value_end_to_end, but the host hands over the DECODED native-size images (%.1f MB "
                                             "per step); float64 resize + normalise + flip pair on the device, bit-identical"
                                             % (wl.u8_offs[-1] / 1e6))
            for k in ("pool", "finisher", "camcopier"):
                wl.e2e[k].shutdown(wait=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if world == 1:
            # BASELINE configs 5 and 4 timed by the same default command: HistoSegNet on ADP-like patches and
            # IRNet inference at VOC size, each through its driver; full detail: --workload hsn / profiles/bench_irn.py
            wl.drain()
            try:
                ha = argparse.Namespace(**vars(args))
                ha.batch, ha.steps, ha.warmup, ha.no_cpu_baseline = 16, 18, 1, True  # (18 batches: one late batch of 9 moved the rate by 20 %)
                hs = hsn_measure(ha, device)
                stages["value_hsn"] = {"value": hs["value"], "unit": "images/s", "ms_per_step": hs["ms_per_step"], "dtype": hs["dtype"],
                                       "batch_images": 16, "workload": hs["config"]["workload"],
                                       "effective_M": hs["config"]["effective_M"]}
            except Exception as e:  # an extra leg must not lose the headline line
                stages["value_hsn"] = {"error": repr(e)}
            try:
                stages["value_irn"] = irn_measure(device, args.precision, reps=6)
            except Exception as e:
                stages["value_irn"] = {"error": repr(e)}
            try:  # BASELINE config 1 through the product driver step.make_cam.run (files on a tmpfs)
                stages["value_make_cam"] = make_cam_measure(device, args.precision)
            except Exception as e:
                stages["value_make_cam"] = {"error": repr(e)}

    if rank == 0:
        images = args.batch * (steps if args.scaling == "strong" else steps * world)
        out = {
            "metric": "images/sec CAM+CRF pseudo-labels, VOC2012 321x321",
            "value": round(images / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / my_steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": "%s CAM + dense-CRF (M=21, 10 mean-field iters), %dx%d, batch %d "
                                   "images (=%d samples) per GPU, image-sharded" % (args.arch, S, S, args.batch,
                                                                                   2 * args.batch)
                       if args.workload == "cam_crf" else
                       "%s CAM (make_cam), %dx%d, batch %d images per GPU" % (args.arch, S, S, args.batch),
                       "batch_images": args.batch, "num_classes": NUM_CLASSES, "crf_config": list(CRF_CFG),
                       "parallelism": "image-sharded x%d, no collective" % world,
                       "steps_per_rank": my_steps,
                       "step_overlap": "mean-field loop of step i overlaps conv stack + lattice build of step i+1 "
                                       "(3 HIP streams)" if pipelined else "none"},
            "roofline": roofline,
            "stages": stages,
            # N = 1 (also `--gpus 1` of a scaling run) is this very path with ranks_seen = 1: no launcher, no process group
            "ranks": ranks_info,
        }
        if steady is not None:
            out["value_steady"] = steady["value"]
            out["steady"] = steady
        if not args.no_cpu_baseline and world == 1 and args.arch == "resnet50":
            labels = None
            if args.workload == "cam_crf":  # the product chain's label maps of this batch, for the parity object
                import numpy as np

                wl.step()
                labels = wl.ctx.to_host(wl.label_dev, (args.batch, S * S), np.int32)
            out["cpu_baseline"], parity = cpu_baseline(wl, args.cpu_seconds, labels)
            if parity is not None:
                out["parity"] = parity
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
