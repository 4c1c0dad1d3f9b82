"""Modified-VGG16 CAM network -- mirror of 03b_irn/net/vgg16_cam.py (CAM.forward :24-60) on
net/vgg16.py:44 + common_cnn.make_layers :128-141.  Weights: a torch-style state_dict with keys
`vgg16.layer<L>.<idx>.{weight,bias,...}`, `vgg16.classifier.0.{weight,bias}` and `thresholds`
(what common_cnn._load_pretrained leaves in the module after transplanting the Keras .h5/.mat,
common_cnn.py:25-82)."""
import numpy as np

from .. import _lib
from .common import DeviceCAMBase

ADP_INDS_X17 = [2, 3, 4, 6, 7, 8, 9, 12, 13, 14, 16, 17, 18, 21, 22, 23, 25, 26, 28, 29, 30, 32, 33, 35, 37, 38, 40,
                45, 48, 49, 50]  # common_cam.py:26-29


class CAM(DeviceCAMBase):
    arch = _lib.ARCH_VGG16_CAM
    root = "vgg16"

    def __init__(self, model_dir=None, dataset="voc12", tag="", num_classes=20, use_cls=None, precision=None):
        super().__init__(num_classes, precision)
        self.model_dir = model_dir
        self.dataset = dataset
        self.tag = tag
        self.use_cls = use_cls
        self.thresholds = 0.5 * np.ones(num_classes, dtype=np.float32)  # common_cnn.py:22
        self.batchnorm = self._has_batchnorm()
        if model_dir is not None and tag:
            self._load_pretrained(model_dir, tag)

    def _has_batchnorm(self):
        return self.dataset not in ("adp_morph", "adp_func")  # vgg16_cam.py:16-19: the ADP VGG16 models carry no BatchNorm

    def _load_pretrained(self, model_dir, tag):
        """CommonCNN._load_pretrained (common_cnn.py:25-41): Keras <tag>.h5 + <tag>.mat under <model_dir>/<tag>/.
        When the files are absent the wrapper stays empty until load_state_dict() is called (the reference would
        raise at construction; make_cam.run passes `args.state_dict` for weightless dry runs)."""
        import os

        from .common import load_pretrained

        if os.path.exists(os.path.join(model_dir, tag, tag + ".h5")):
            self.load_state_dict(load_pretrained(model_dir, tag, self.root, self.batchnorm), strict=True)

    def load_state_dict(self, state_dict, strict=True):
        super().load_state_dict(state_dict, strict)
        if "thresholds" in self._sd:
            self.thresholds = np.asarray(self._sd.pop("thresholds"), dtype=np.float32)
        return self

    def predict_labels(self, score):
        """vgg16_cam.py:37-45: score >= thresholds; if nothing passes on non-ADP data force argmax."""
        y = score >= self.thresholds[:len(score)]
        if self.dataset not in ("adp_morph", "adp_func") and y.sum() == 0:
            y[np.argmax(score)] = True
        if "X1.7" in self.tag:
            y = y[ADP_INDS_X17]
        return y

    # ---- ADP background / other-tissue channels: common_cam.py:31-92, on the device --------------------------
    def adp_channel_lists(self):
        """(mode, use, adipose, exceptions) as channel indices of the network's CAM stack: mode 0 = adp_morph
        (_adp_modify_morph :31-55), 1 = adp_func (_adp_modify_func :57-92); the X1.7 class filter (:26-29, applied to the
        CAM before the modification, vgg16_cam.py:49-50) is composed into the lists."""
        filt = ADP_INDS_X17 if "X1.7" in self.tag else list(range(self.num_classes))
        use = [filt[int(i)] for i in self.use_cls]
        return (0 if self.dataset == "adp_morph" else 1), use, [filt[i] for i in (18, 19, 20)], [filt[i] for i in (28, 29, 30)]

    def adp_out_channels(self):
        return len(self.use_cls) + (1 if self.dataset == "adp_morph" else 2)

    def adp_modify_device(self, ctx, cam_dev, B, n_sc, h, w, origs, out_dev=None, stage=None):
        """The ADP map stacks of a device batch (vgg16_cam.py:51-58), never leaving the GPU:
        0.75 * expit(4 * (mean_rgb - 240)) of every ORIGINAL (un-normalised, un-flipped) image, Gaussian smoothed (sigma 2),
        cv2-bilinear resized to the CAM grid (wsc_hsn_background), then the background / 'other' channels joined with the
        use_cls CAM channels and summed over the scales (wsc_cam_adp_modify).
          cam_dev float32 [B][n_sc][C][h][w];  origs: B * n_sc uint8 (H0, W0, 3) arrays in (image, scale) order;
          stage: optional (PinnedBuffer, DeviceBuffer) pair of >= sum(H0 W0 3) bytes for an asynchronous upload.
        -> (out_dev float32 [B][Cout][h][w], Cout)."""
        origs = [np.ascontiguousarray(o, dtype=np.uint8) for o in origs]
        assert len(origs) == B * n_sc and all(o.ndim == 3 and o.shape[2] == 3 for o in origs)
        offs = np.concatenate(([0], np.cumsum([o.size for o in origs]))).astype(np.int64)
        total = int(offs[-1])
        if stage is not None:
            pin, rgb_dev = stage
            buf = pin.view((total,), np.uint8)
            for k, o in enumerate(origs):
                buf[offs[k]:offs[k + 1]] = o.reshape(-1)
            ctx.h2d_async(rgb_dev, pin, total)
        else:
            rgb_dev = ctx.to_device(np.concatenate([o.reshape(-1) for o in origs]), pooled=True)
        hw = h * w
        bg_dev = ctx.alloc(B * n_sc * hw * 8, pooled=True)
        k = 0
        while k < len(origs):  # runs of equal size (ADP patches all are 272 x 272: one call)
            e = k + 1
            while e < len(origs) and origs[e].shape == origs[k].shape:
                e += 1
            H0, W0 = origs[k].shape[:2]
            _lib.hsn_background(ctx, rgb_dev.ptr + int(offs[k]), e - k, H0, W0, bg_dev.ptr + k * hw * 8, out_hw=(h, w))
            k = e
        mode, use, adip, exc = self.adp_channel_lists()
        Cout = len(use) + 1 + mode
        if out_dev is None:
            out_dev = ctx.alloc(B * Cout * hw * 4, pooled=True)
        _lib.cam_adp_modify(ctx, cam_dev, B, n_sc, self.num_classes, hw, bg_dev, mode, use, adip, exc, out_dev)
        bg_dev.free()  # pooled: reused in stream order
        if stage is None:
            rgb_dev.free()
        return out_dev, Cout

    def forward(self, x, x_orig=None):
        """-> (cam (C,h,w), y bool (C,)) as vgg16_cam.py:24-60 / m7_cam.py:22-57; for the ADP datasets
        `x_orig` is the (2,H0,W0,3) uint8 image pair of the ADP dataloader."""
        assert (self.dataset in ("adp_morph", "adp_func")) ^ (x_orig is None)
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        if x_orig is None:
            cam, score = self.forward_batch(xn[None], want_score=True)
            cam = cam[0]
            if "X1.7" in self.tag:
                cam = cam[ADP_INDS_X17]
        else:  # the ADP channels are synthesised on the device, between the CAM head and the copy out
            x_orig = x_orig.detach().cpu().numpy() if hasattr(x_orig, "detach") else np.asarray(x_orig)
            net = self._ensure_net()
            ctx = self._ctx
            xc = np.ascontiguousarray(xn[None], dtype=np.float32)
            S = xc.shape[-1]
            h = net.cam_size(S)
            x_dev = ctx.to_device(xc, pooled=True)
            cam_dev = ctx.alloc(self.num_classes * h * h * 4, pooled=True)
            score_dev = ctx.alloc(self.num_classes * 4, pooled=True)
            net.forward_cam(x_dev, 1, S, cam_dev, score_dev)
            out_dev, Cout = self.adp_modify_device(ctx, cam_dev, 1, 1, h, h, [x_orig[0]])
            cam = ctx.to_host(out_dev, (Cout, h, h), np.float32)
            score = ctx.to_host(score_dev, (1, self.num_classes), np.float32)
            for b in (x_dev, cam_dev, score_dev, out_dev):
                b.free()
        y = self.predict_labels(score[0])
        if is_torch:
            import torch

            return torch.from_numpy(cam), torch.from_numpy(y)
        return cam, y

    __call__ = forward
