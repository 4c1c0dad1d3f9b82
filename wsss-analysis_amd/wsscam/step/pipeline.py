"""Host pipeline of the make_cam worker (03b_irn/step/make_cam.py:25-93 is a serial loop: DataLoader item ->
.cuda() -> model -> two interpolates -> .cpu() x3 -> np.save, one image at a time).

Here a worker keeps the GPU fed from several directions at once:
  loader threads   dataset[i] (decode, resize, normalise) written straight into a lane's page-locked input buffer
  lanes            n_lanes x (wsc_ctx = one HIP stream, pinned input + output staging, device buffers): batch i runs on
                   lane i % n_lanes -- pinned H2D, conv stack, CAM head, make_cam tail, pinned D2H, all asynchronous
  finisher threads wait for a lane's stream (wsc_sync releases the GIL), then hand the images to the
  writer threads   np.save of the pickled {"keys", "cam", "high_res"} dicts (the reference's format, unchanged)
Per-image outputs stay idempotent: a rerun rewrites the same files.  No collective, no shared state between GPUs.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _lib


def host_thread_budget(world, cores=None):
    """Loader / writer threads of ONE worker when `world` workers (one per GPU) share the host: the cores this process may
    run on (sched_getaffinity) divided by the workers, 5/8 of them decoding + resizing (loaders), the rest pickling + writing
    (writers), each capped at 8 -- a single worker on a 128-core host keeps round 3's 8 + 8; 8 workers on a 64-core host get
    5 + 3 each instead of 8 x 16 threads on 64 cores.  (The lanes' finisher threads sleep in wsc_sync and cost no core.)"""
    if cores is None:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
    per = max(1, int(cores) // max(1, int(world)))
    loaders = max(1, min(8, (per * 5) // 8))
    writers = max(1, min(8, per - loaders))
    return {"cores": int(cores), "cores_per_rank": per, "n_loaders": loaders, "n_writers": writers}


class _Lane:
    def __init__(self, device, B, S, C, h, n_sc=1, adp_channels=0):
        self.ctx = _lib.Context(device)
        self.B, self.S, self.C, self.h, self.n_sc = B, S, C, h, n_sc
        self.in_bytes = B * n_sc * 2 * 3 * S * S * 4
        self.pin_in = self.ctx.host_alloc(self.in_bytes)
        self.x_view = self.pin_in.view((B, n_sc, 2, 3, S, S), np.float32)  # the scales of an image are consecutive samples
        self.x_dev = self.ctx.alloc(self.in_bytes)
        self.cam_dev = self.ctx.alloc(B * n_sc * C * h * h * 4)
        self.sum_dev = self.ctx.alloc(B * C * h * h * 4) if n_sc > 1 else None
        # ADP datasets: the stack [background | (other) | use_cls channels] of wsc_cam_adp_modify, summed over the scales
        self.adp_dev = self.ctx.alloc(B * adp_channels * h * h * 4) if adp_channels else None
        self.score_dev = self.ctx.alloc(B * n_sc * C * 4)
        self.pin_score = self.ctx.host_alloc(B * n_sc * C * 4)
        self.out_cap = 0
        self.pin_out = None
        self.s_dev = self.h_dev = None
        self.u8_cap = 0
        self.pin_u8 = self.u8_dev = None
        self.free = threading.Event()
        self.free.set()

    def ensure_out(self, s_tot, h_tot):
        need = (max(s_tot, 1) + max(h_tot, 1)) * 4
        if need > self.out_cap:
            cap = int(need * 1.5) + (1 << 20)
            for b in (self.pin_out, self.s_dev, self.h_dev):
                if b is not None:
                    b.free()
            self.pin_out = self.ctx.host_alloc(cap)
            self.s_dev = self.ctx.alloc(cap)
            self.h_dev = self.ctx.alloc(cap)
            self.out_cap = cap

    def ensure_u8(self, nbytes):
        if nbytes > self.u8_cap:
            cap = int(nbytes * 1.5) + (1 << 20)
            for b in (self.pin_u8, self.u8_dev):
                if b is not None:
                    b.free()
            self.pin_u8 = self.ctx.host_alloc(cap)
            self.u8_dev = self.ctx.alloc(cap)
            self.u8_cap = cap

    def close(self):
        self.ctx.sync()
        for b in (self.pin_in, self.pin_out, self.pin_score, self.x_dev, self.cam_dev, self.sum_dev, self.adp_dev, self.score_dev, self.s_dev,
                  self.h_dev, self.pin_u8, self.u8_dev):
            if b is not None:
                b.free()
        self.ctx.close()


class CamPipeline:
    """Batched, overlapped make_cam worker for one GPU.  `keys_fn(pack, score_row_or_None) -> int64 keys`,
    `save_fn(name, keys, strided, highres)` are the driver's own (make_cam._valid_cat / _save)."""

    def __init__(self, model, device, batch_images, S, keys_fn, save_fn, needs_score, n_lanes=3, n_loaders=None, n_writers=None,
                 norm=None, n_scales=1, world=1, chain_stacks=True):
        """n_loaders / n_writers None: sized from the host cores this worker may use when `world` workers share the host
        (host_thread_budget).  chain_stacks: the lanes' conv stacks run one after the other on the device (_device_step)."""
        budget = host_thread_budget(world)
        n_loaders = budget["n_loaders"] if n_loaders is None else int(n_loaders)
        n_writers = budget["n_writers"] if n_writers is None else int(n_writers)
        self.n_loaders, self.n_writers, self.world = n_loaders, n_writers, int(world)
        self.model, self.device, self.B, self.S = model, device, batch_images, S
        self.norm = norm  # TorchvisionNormalize of the dataset (device transform of "img_u8" items)
        self.keys_fn, self.save_fn, self.needs_score = keys_fn, save_fn, needs_score
        self.C = model.num_classes
        self.h = model.cam_size(S)
        self.n_sc = int(n_scales)  # args.cam_scales: every image contributes n_sc network inputs, their CAMs are summed
        # ADP datasets (vgg16_cam.py:51-58): background / 'other' channels from the original images, on the lane's stream
        self.adp = getattr(model, "dataset", None) in ("adp_morph", "adp_func")
        self.lanes = [self._make_lane(device, batch_images, S) for _ in range(n_lanes)]
        self.loaders = ThreadPoolExecutor(n_loaders, thread_name_prefix="wsc-load")
        self.writers = ThreadPoolExecutor(n_writers, thread_name_prefix="wsc-save")
        self.finishers = ThreadPoolExecutor(n_lanes, thread_name_prefix="wsc-finish")
        self.errors = []
        self.images_done = 0
        self.chain = _lib.StackChain() if chain_stacks else None  # (see _device_step)

    def _make_lane(self, device, batch_images, S):
        return _Lane(device, batch_images, S, self.C, self.h, self.n_sc, self.model.adp_out_channels() if self.adp else 0)

    # -- stages --------------------------------------------------------------------------------------------
    def _load_into(self, lane, k, dataset, idx):
        pack = dataset[idx]
        if "img" not in pack:  # device transform: the decoded uint8 image travels, wsc_msf_input_u8 does the rest
            return dict(pack)
        v = pack["img"]
        img = np.stack([np.asarray(a, dtype=np.float32) for a in (v if isinstance(v, (list, tuple)) else [v])])
        if img.shape != lane.x_view.shape[1:]:
            raise ValueError("make_cam: network inputs must be %d x (2, 3, %d, %d) for every image of a run; %s has %s"
                             % (self.n_sc, self.S, self.S, pack.get("name"), img.shape))
        lane.x_view[k] = img  # page-locked: the H2D below is a straight DMA
        return {key: v for key, v in pack.items() if key != "img"}

    def _copy_out(self, lane, s_tot, h_tot):
        """cam / high_res of a finished batch -> the lane's page-locked buffer.  Issued by the finisher AFTER it has seen the
        lane's kernels finish, not queued behind them: the runtime turns a queued copy's dependency into a poll command on the
        DMA engine's own in-order queue, where it holds back every copy submitted after it -- the next lane's batch upload
        included -- until this lane's conv stack is done (measured on the bench's end-to-end leg, round 6: 1.4 ms per step)."""
        ctx = lane.ctx
        ctx.d2h_async(lane.pin_out, lane.s_dev, max(s_tot, 1) * 4)
        ctx.d2h_async(lane.pin_out, lane.h_dev, max(h_tot, 1) * 4, dst_offset=max(s_tot, 1) * 4)
        ctx.sync()

    def _finish(self, lane, metas, keys, shapes, s_off, h_off, s_tot, h_tot):
        try:
            lane.ctx.sync()  # this lane's stream only; the other lanes keep running
            self._copy_out(lane, s_tot, h_tot)
            strided = lane.pin_out.view((max(s_tot, 1),), np.float32)
            highres = lane.pin_out.view((lane.out_cap // 4 - max(s_tot, 1),), np.float32, offset_bytes=max(s_tot, 1) * 4)
            futs = []
            for b, meta in enumerate(metas):
                K, h4, w4, H0, W0 = shapes[b]
                sc = strided[s_off[b]:s_off[b] + K * h4 * w4].reshape(K, h4, w4)
                hc = highres[h_off[b]:h_off[b] + K * H0 * W0].reshape(K, H0, W0)
                futs.append(self.writers.submit(self.save_fn, meta["name"], keys[b], sc, hc))
            for f in futs:
                f.result()
            self.images_done += len(metas)
        except Exception as e:  # surfaced by run()
            self.errors.append(e)
        finally:
            lane.free.set()

    def _start(self, lane, dataset, indices):
        n = len(indices)
        metas = [f.result() for f in [self.loaders.submit(self._load_into, lane, k, dataset, i) for k, i in enumerate(indices)]]
        keys, shapes, s_off, h_off, s_tot = self._device_step(lane, n, metas)
        h_tot = sum(K * H0 * W0 for K, _, _, H0, W0 in shapes)
        lane.free.clear()
        self.finishers.submit(self._finish, lane, metas, keys, shapes, s_off, h_off, s_tot, h_tot)

    def _device_step(self, lane, n, metas):
        """Everything a batch does on its lane's stream (H2D, conv stack, CAM head, tail -- all asynchronous; the finisher copies out): returns what
        the finisher needs to cut the outputs out of the lane's staging buffer.  (The CPU tests of the N-worker host side
        replace this method and _make_lane; nothing else of the pipeline touches the device.)"""
        ctx = lane.ctx
        if "img_u8" in metas[0]:
            # decoded images: (sum H0 W0 3) bytes over PCIe instead of n x 2.47 MB of float32; resize + normalise + flip
            # pair on the device, bit-identical to the host transform (csrc/input.hip)
            imgs_u8 = []
            for m in metas:  # (image, scale) order = sample order of the device batch
                v = m.pop("img_u8")
                imgs_u8.extend(v if isinstance(v, (list, tuple)) else [v])
            sizes_u8 = [tuple(int(v) for v in np.asarray(a).shape[:2]) for a in imgs_u8]
            offs = np.concatenate(([0], np.cumsum([h * w * 3 for h, w in sizes_u8]))).astype(np.int64)
            lane.ensure_u8(int(offs[-1]))
            buf = lane.pin_u8.view((lane.u8_cap,), np.uint8)
            for k, a in enumerate(imgs_u8):
                buf[offs[k]:offs[k + 1]] = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
            ctx.h2d_async(lane.u8_dev, lane.pin_u8, int(offs[-1]))
            norm = self.norm
            _lib.msf_input_u8(ctx, lane.u8_dev, sizes_u8, offs[:-1], self.S, norm.mean, norm.std, lane.x_dev,
                              pre_div255=norm.norm_mode == "float", pair=True)
        else:
            ctx.h2d_async(lane.x_dev, lane.pin_in, n * self.n_sc * 2 * 3 * self.S * self.S * 4)
        # One conv stack at a time (_lib.StackChain): the stack of batch i waits on the device for the stack of batch i-1, which
        # ran on another lane; the uploads, the input transform, the tails and the copy-outs of the other lanes still overlap it
        net = self.model._ensure_net()

        def stack():
            net.forward_cam(lane.x_dev, n * self.n_sc, self.S, lane.cam_dev, lane.score_dev if self.needs_score else None, ctx=ctx)

        if self.chain is not None:
            self.chain.run(ctx, stack)
        else:
            stack()
        cam_dev, C = lane.cam_dev, self.C
        if self.adp:
            # common_cam.py:31-92 on the device: the original (un-flipped) image of every scale travels as uint8 through the
            # lane's pinned staging, wsc_hsn_background + wsc_cam_adp_modify build the stack (summed over the scales)
            origs = []
            for m in metas:
                v = m.pop("orig_img")
                origs.extend(np.asarray(a)[0] for a in (v if isinstance(v, (list, tuple)) else [v]))
            lane.ensure_u8(sum(int(o.size) for o in origs))
            cam_dev, C = self.model.adp_modify_device(ctx, lane.cam_dev, n, self.n_sc, self.h, self.h, origs, out_dev=lane.adp_dev,
                                                      stage=(lane.pin_u8, lane.u8_dev))
        elif self.n_sc > 1:  # make_cam.py:62-69: sum over the scales (all of one size: see wsc_cam_sum_scales)
            _lib.cam_sum_scales(ctx, lane.cam_dev, n, self.n_sc, self.C * self.h * self.h, lane.sum_dev)
            cam_dev = lane.sum_dev
        score = None
        if self.needs_score:  # predicted labels decide which maps are produced: one small read-back
            ctx.sync()  # (first the kernels, then the copy: see _copy_out)
            ctx.d2h_async(lane.pin_score, lane.score_dev, n * self.n_sc * self.C * 4)
            ctx.sync()
            score = lane.pin_score.view((n, self.n_sc, self.C), np.float32)[:, 0].copy()  # labels[0]: the first scale's
        keys = [self.keys_fn(m, None if score is None else score[b]) for b, m in enumerate(metas)]
        sizes = [tuple(int(v) for v in m["size"]) for m in metas]
        s_tot = sum(len(k) * ((H - 1) // 4 + 1) * ((W - 1) // 4 + 1) for k, (H, W) in zip(keys, sizes))
        h_tot = sum(len(k) * H * W for k, (H, W) in zip(keys, sizes))
        lane.ensure_out(s_tot, h_tot)
        _, _, s_off, h_off, shapes = _lib.cam_postprocess(ctx, cam_dev, n, C, self.h, self.h, sizes, keys,
                                                          lane.s_dev, lane.h_dev)
        return keys, shapes, s_off, h_off, s_tot  # (the copy-out is the finisher's: _copy_out)

    # -- driver --------------------------------------------------------------------------------------------
    def run(self, dataset, indices=None):
        idx = list(range(len(dataset))) if indices is None else list(indices)
        for bi, i0 in enumerate(range(0, len(idx), self.B)):
            lane = self.lanes[bi % len(self.lanes)]
            lane.free.wait()
            if self.errors:
                break
            self._start(lane, dataset, idx[i0:i0 + self.B])
        for lane in self.lanes:
            lane.free.wait()
        if self.errors:
            raise self.errors[0]

    def close(self):
        for ex in (self.loaders, self.writers, self.finishers):
            ex.shutdown(wait=True)
        for lane in self.lanes:
            lane.close()
