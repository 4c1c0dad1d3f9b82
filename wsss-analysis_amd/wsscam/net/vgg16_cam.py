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

    # ---- ADP background / other-tissue channels: common_cam.py:31-92 ------------------------------------
    def _adp_background(self, cam, img_orig):
        """0.75 * sigmoid(4 * (mean_rgb - 240)) of the ORIGINAL (un-normalised, un-flipped) image, Gaussian
        smoothed (sigma 2), resized to the CAM size (cv2.resize bilinear)."""
        import scipy.ndimage
        import scipy.special

        from ..cues.utilities import resize_stack

        mean_img = np.mean(np.asarray(img_orig[0], dtype=np.float32), axis=2)
        bg = 0.75 * scipy.special.expit(4 * (mean_img - 240))
        bg = scipy.ndimage.gaussian_filter(bg, sigma=2)
        if bg.shape != cam.shape[1:]:
            bg = resize_stack(bg[None, None], cam.shape[1:], ctx=self.ctx)[0, 0]
        return bg.astype(np.float32)

    def _adp_modify_morph(self, cam, img_orig):
        """common_cam.py:31-55: background = relu(bg - max adipose CAM), prepended to cam[use_cls]."""
        bg = self._adp_background(cam, img_orig)
        background = np.maximum(bg - np.max(cam[[18, 19, 20]], axis=0), 0)
        return np.concatenate((background[None], cam[self.use_cls]), axis=0)

    def _adp_modify_func(self, cam, img_orig):
        """common_cam.py:57-92: background = bg - max exception CAM (no relu), then an 'other' channel
        max(0.05 * (1 - max_c modified), max adipose CAM) inserted after it."""
        bg = self._adp_background(cam, img_orig)
        background = bg - np.max(cam[[28, 29, 30]], axis=0)
        modified = np.concatenate((background[None], cam[self.use_cls]), axis=0)
        other = np.maximum(0.05 * (1 - np.max(modified, axis=0)), np.max(cam[[18, 19, 20]], axis=0))
        return np.concatenate((modified[:1], other[None], modified[1:]), axis=0)

    def adp_modify(self, cam, img_orig):
        """cam (C,h,w) of one image -> the ADP map stack make_cam post-processes (vgg16_cam.py:51-58)."""
        if "X1.7" in self.tag:
            cam = cam[ADP_INDS_X17]
        if self.dataset == "adp_morph":
            return self._adp_modify_morph(cam, img_orig)
        if self.dataset == "adp_func":
            return self._adp_modify_func(cam, img_orig)
        return cam

    def forward(self, x, x_orig=None):
        """-> (cam (C,h,w), y bool (C,)) as vgg16_cam.py:24-60 / m7_cam.py:22-57; for the ADP datasets
        `x_orig` is the (2,H0,W0,3) uint8 image pair of the ADP dataloader."""
        assert (self.dataset in ("adp_morph", "adp_func")) ^ (x_orig is None)
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        cam, score = self.forward_batch(xn[None], want_score=True)
        cam, y = cam[0], self.predict_labels(score[0])
        if x_orig is not None:
            x_orig = x_orig.detach().cpu().numpy() if hasattr(x_orig, "detach") else np.asarray(x_orig)
        cam = self.adp_modify(cam, x_orig)
        if is_torch:
            import torch

            return torch.from_numpy(cam), torch.from_numpy(y)
        return cam, y

    __call__ = forward
