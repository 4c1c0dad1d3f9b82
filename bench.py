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
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0}
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per step per GPU")
    ap.add_argument("--precision", default="f16", choices=["bf16", "f16", "bf16x3"])
    ap.add_argument("--workload", default="cam_crf", choices=["cam_crf", "cam"])
    ap.add_argument("--arch", default="resnet50", choices=["resnet50", "vgg16", "m7"],
                    help="CAM network (resnet50 is the BASELINE.json configuration; vgg16 / m7 are extra "
                         "measurements of the other conv stacks of the reference, 03b_irn/net/{vgg16,m7}.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


class Workload:
    def __init__(self, device, batch, precision, workload, seed, arch="resnet50"):
        import numpy as np

        from oracle import cnn_ref  # synthetic-input generator only (no compute of the product path)
        from wsscam import _lib

        global S, GFLOP_PER_IMAGE
        S = INPUT_SIZE_BY_ARCH[arch]
        GFLOP_PER_IMAGE = GFLOP_PER_IMAGE_BY_ARCH[arch]

        self.np, self._lib = np, _lib
        self.B, self.workload = batch, workload
        self.ctx = _lib.Context(device)
        # second context (own stream) for the lattice build: it needs only the RGB images, so it runs
        # concurrently with the CNN forward pass of the same batch and joins before the inference
        self.ctx_build = _lib.Context(device)
        # third context: the mean-field loop.  With --pipeline (default) step i's loop runs here while
        # step i+1's conv stack (self.ctx) and lattice build (self.ctx_build) are already under way.
        self.ctx_crf = _lib.Context(device)
        self.pending = None  # lattices of the step whose mean-field loop is still in flight
        prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3}[precision]
        if arch == "resnet50":
            sd = {k: v.numpy() for k, v in cnn_ref.make_resnet50_cam_state_dict(NUM_CLASSES, seed=0).items()}
            self.net = _lib.Net(self.ctx, _lib.ARCH_RESNET50_CAM, sd, NUM_CLASSES, prec)
        elif arch == "vgg16":
            sd = {k: v.numpy() for k, v in cnn_ref.make_plain_state_dict("vgg16", cnn_ref.VGG16_CFG, NUM_CLASSES, True,
                                                                       seed=0).items()}
            self.net = _lib.Net(self.ctx, _lib.ARCH_VGG16_CAM, sd, NUM_CLASSES, prec)
        else:
            sd = {k: v.numpy() for k, v in cnn_ref.make_plain_state_dict("m7", cnn_ref.M7_CFG, NUM_CLASSES, True,
                                                                       seed=0).items()}
            rngw = np.random.default_rng(5)
            sd["gradcam_weights"] = rngw.normal(0, 0.05, (256, NUM_CLASSES)).astype(np.float32)
            self.net = _lib.Net(self.ctx, _lib.ARCH_M7_CAM, sd, NUM_CLASSES, prec)
        self.arch = arch
        self.h = self.net.cam_size(S)
        rng = np.random.default_rng(20121 + seed)
        gold = np.load(os.path.join(ROOT, "tests", "golden", "resnet50_cam.npz"))
        labels = gold["trainaug_labels"]
        xs, rgbs, self.sizes, self.keys = [], [], [], []
        for i in range(batch):
            H0, W0 = cnn_ref.VOC_SIZES[(i + seed) % len(cnn_ref.VOC_SIZES)]
            img = cnn_ref.synth_image(rng, H0, W0)
            r = cnn_ref.resize_bilinear_f64(img, (S, S))
            x = np.transpose(cnn_ref.normalize_int(r), (2, 0, 1))
            xs.append(np.stack([x, np.flip(x, -1)], 0))
            rgbs.append(np.clip(np.rint(r), 0, 255).astype(np.uint8))
            self.sizes.append((H0, W0))
            self.keys.append(np.nonzero(labels[(i + seed * batch) % len(labels)])[0].astype(np.int32))
        self.x_host = np.ascontiguousarray(np.stack(xs), dtype=np.float32)
        self.rgb_host = np.ascontiguousarray(np.stack(rgbs))
        ctx = self.ctx
        self.x_dev = ctx.to_device(self.x_host)
        self.rgb_dev = ctx.to_device(self.rgb_host)
        self.cam_dev = ctx.alloc(batch * NUM_CLASSES * self.h * self.h * 4)
        # native-size make_cam outputs
        s_tot = sum(len(k) * ((H - 1) // 4 + 1) * ((W - 1) // 4 + 1) for k, (H, W) in zip(self.keys, self.sizes))
        h_tot = sum(len(k) * H * W for k, (H, W) in zip(self.keys, self.sizes))
        self.strided_dev = ctx.alloc(max(s_tot, 1) * 4)
        self.highres_dev = ctx.alloc(max(h_tot, 1) * 4)
        # CRF stack
        N = S * S
        # unaries / labels are double-buffered: step i+1 writes its unaries while step i's loop reads its own
        self.unary_bufs = [ctx.alloc(batch * (NUM_CLASSES + 1) * N * 4) for _ in range(2)]
        self.label_bufs = [ctx.alloc(batch * N * 4) for _ in range(2)]
        self.unary_dev, self.label_dev = self.unary_bufs[0], self.label_bufs[0]
        self.parity = 0
        self.vg = self.vb = None

    # -- pieces ------------------------------------------------------------------------------
    def run_cnn(self):
        self.net.forward_cam(self.x_dev, self.B, S, self.cam_dev, None)

    def run_tail(self):
        self._lib.cam_postprocess(self.ctx, self.cam_dev, self.B, NUM_CLASSES, self.h, self.h, self.sizes, self.keys,
                                  self.strided_dev, self.highres_dev)

    def run_unary(self):
        # upsample all 20 class maps to S x S, x /= max + 1e-5, [bg = 0.15 | maps] -> unaries: one fused call
        # (wsc_cam_postprocess + wsc_unary_from_maps give the same bits through 0.5 GB of intermediate maps)
        self._lib.cam_unary(self.ctx, self.cam_dev, self.B, NUM_CLASSES, self.h, self.h, S, S, 0.15, self.unary_dev)

    def crf_create(self):
        return self._lib.Crf(self.ctx_build, self.rgb_dev, self.B, S, S, CRF_CFG[0], CRF_CFG[2], CRF_CFG[3])

    def crf_infer(self, crf, ctx=None):
        crf.inference(self.unary_dev, NUM_CLASSES + 1, CRF_CFG[1], CRF_CFG[4], CRF_CFG[5], None, self.label_dev,
                      ctx=ctx or self.ctx)

    def drain(self):
        """Wait for the mean-field loop still in flight (pipelined mode) and release its lattices."""
        if self.pending is not None:
            self.ctx_crf.sync()
            self.pending.close()
            self.pending = None

    def step_pipelined(self):
        """Same work per step as step(), issued so that consecutive steps overlap:
        stream A (ctx)       conv stack, tail, unaries of step i
        stream B (ctx_build) lattice build of step i
        stream C (ctx_crf)   mean-field loop of step i, after A and B  -- still running while the host
                             already enqueues step i+1 on A and B."""
        self.parity ^= 1
        self.unary_dev, self.label_dev = self.unary_bufs[self.parity], self.label_bufs[self.parity]
        self.run_cnn()
        crf = self.crf_create()          # host blocks on the build stream only; A and C keep running
        if self.vg is None:
            self.vg, self.vb = crf.lattice_sizes()
        self.run_tail()
        self.run_unary()
        self.drain()                     # step i-1's loop must be done before its workspace is reused
        self.ctx_crf.wait_for(self.ctx)
        self.ctx_crf.wait_for(self.ctx_build)
        self.crf_infer(crf, ctx=self.ctx_crf)
        self.pending = crf

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

    def timed(self, fn, reps):
        """Average device time of fn() over reps, HIP events on the ctx stream."""
        fn()  # untimed: the first call on this ctx may grow its workspace arena (a hipMalloc of a few GB)
        self.ctx.sync()
        self.ctx.timer_begin()
        for _ in range(reps):
            fn()
        return self.ctx.timer_end() / reps


def cpu_baseline(wl, budget_s):
    """Oracle on the host cores: torch-CPU fp32 restatement of make_cam._work (batch = 1 image, as the
    reference runs it) + the single-threaded C dense-CRF restatement (pydensecrf is single-threaded).
    kind = "port": the reference itself cannot travel to the GPU box."""
    import numpy as np
    import torch

    from oracle import cnn_ref
    from tests import helpers

    sd = cnn_ref.make_resnet50_cam_state_dict(NUM_CLASSES, seed=0)
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
    t_budget = budget_s * 0.5
    while n_cam < wl.B and (n_cam < 2 or t_cam < t_budget):
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
    n_crf, t_crf = 0, 0.0
    if wl.workload == "cam_crf":
        while n_crf < n_cam and (n_crf < 1 or t_crf < budget_s * 0.5):
            t0 = time.perf_counter()
            with torch.no_grad():
                _, hi = cnn_ref.make_cam_tail(cams[n_crf], (S, S), torch.arange(NUM_CLASSES))
            v = np.concatenate([np.full((1, S * S), 0.15, np.float32), hi.numpy().reshape(NUM_CLASSES, -1)], 0)
            p = v / v.sum(0, keepdims=True)
            U = -np.log(np.clip(p, 1e-5, 1.0)).astype(np.float32)
            helpers.crf_oracle(wl.rgb_host[n_crf], U, CRF_CFG)
            t_crf += time.perf_counter() - t0
            n_crf += 1
    per_img = t_cam / n_cam + (t_crf / n_crf if n_crf else 0.0)
    return {"value": round(1.0 / per_img, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d images torch-CPU fp32 ResNet50-CAM+tail (%.2f s/img, %d threads) + %d images C dense-CRF "
                      "M=21 T=10 (%.2f s/img, 1 thread)" % (n_cam, t_cam / n_cam, cores, n_crf,
                                                            t_crf / n_crf if n_crf else 0.0)}


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

    wl = Workload(device, args.batch, args.precision, args.workload, seed=rank, arch=args.arch)

    def barrier():
        wl.ctx.sync()
        wl.ctx_build.sync()
        wl.ctx_crf.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    pipelined = args.workload == "cam_crf" and not args.no_pipeline
    do_step = wl.step_pipelined if pipelined else wl.step
    for _ in range(args.warmup):
        do_step()
    wl.drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        do_step()
    wl.drain()
    wl.ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

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
    kernels = {}
    for name, (calls, ms, work) in prof.items():
        is_conv = name.startswith("conv_igemm")
        rate = work / (ms * 1e-3) / (1e12 if is_conv else 1e9) if ms > 0 else 0.0
        kernels[name] = {"launches_per_step": calls // 2, "avg_us": round(ms / calls * 1e3, 2),
                         "ms_per_step": round(ms / 2, 4), ("TFLOP/s" if is_conv else "GB/s"): round(rate, 2)}
    dom = max((n for n in prof if n != "crf_build(all)"), key=lambda n: prof[n][1])
    calls, ms, work = prof[dom]
    if dom.startswith("conv_igemm"):
        roofline = {"bound": "mfma", "achieved": round(work / (ms * 1e-3) / 1e12, 2), "peak": PEAK_TFLOPS[args.precision],
                    "unit": "TFLOP/s", "traffic": None}
    else:
        roofline = {"bound": "hbm", "achieved": round(work / (ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "traffic": None}
    roofline["frac"] = round(roofline["achieved"] / roofline["peak"], 4)
    roofline["kernel"] = dom
    roofline["algorithmic_bytes_per_launch" if roofline["bound"] == "hbm" else "flop_per_launch"] = round(work / calls)
    # HBM bytes per launch of that kernel class from the committed PMC passes (profiles/collect.sh ->
    # profiles/hbm_traffic.json; FETCH_SIZE / WRITE_SIZE cannot be read from inside the process).  Only the
    # default workload is profiled there; anything else reports null.
    tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    default_wl = (args.workload == "cam_crf" and args.arch == "resnet50" and args.batch == 32 and args.precision == "f16")
    if default_wl and os.path.exists(tj):
        with open(tj) as fh:
            cls = json.load(fh).get("classes", {}).get(dom)
        if cls:
            roofline["traffic"] = cls["bytes_per_launch"]
            roofline["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
    roofline["avg_launch_us"] = round(ms / calls * 1e3, 2)
    roofline["launches_per_step"] = calls // 2
    stages["kernels"] = kernels

    if rank == 0:
        images = args.batch * args.steps * world
        out = {
            "metric": "images/sec CAM+CRF pseudo-labels, VOC2012 321x321",
            "value": round(images / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
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
                       "step_overlap": "mean-field loop of step i overlaps conv stack + lattice build of step i+1 "
                                       "(3 HIP streams)" if pipelined else "none"},
            "roofline": roofline,
            "stages": stages,
        }
        if not args.no_cpu_baseline and world == 1 and args.arch == "resnet50":
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
