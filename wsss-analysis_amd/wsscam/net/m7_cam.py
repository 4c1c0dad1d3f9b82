"""M7 Grad-CAM network -- mirror of 03b_irn/net/m7_cam.py (CAM.forward :22-57) on net/m7.py:41.
The 1x1 head is the Grad-CAM weight matrix (F x C) computed once per model on a zeros image
(common_cnn.gen_gradcam_weights :84-121, run by MNet.__init__ through `_load_pretrained(gen_gradcam=True)`,
net/m7.py:22).  A state dict without `gradcam_weights` gets them here the same way -- from the closed form of
net.common.grad_cam_alpha on the Keras model's 224 x 224 input (56 x 56 x 256 final activation, gradient taken through
the last BatchNorm).  The torch port then applies alpha to layer3_p1's output, the POST-BatchNorm map (m7_cam.py:26-28,
45-46), which is what the device head does here."""
import numpy as np

from .. import _lib
from . import vgg16_cam
from .common import grad_cam_alpha, last_bn_affine


class CAM(vgg16_cam.CAM):
    arch = _lib.ARCH_M7_CAM
    root = "m7"
    keras_input_size = 224  # 02_cues/demo.py:60-66, 03b_irn/func_sample.py:151-156

    def __init__(self, model_dir=None, dataset="voc12", tag="", num_classes=20, use_cls=None, precision=None):
        if dataset in ("adp_morph", "adp_func") and "X1.7" in tag:
            num_classes = 51  # m7_cam.py:16-17: the X1.7 models score all 51 ADP classes, filtered to 31 afterwards
        super().__init__(model_dir, dataset, tag, num_classes, use_cls, precision)

    def _has_batchnorm(self):
        return True  # m7_cam.py:18: m7.m7(..., batchnorm=True) for every dataset (only vgg16_cam switches it off for ADP)

    def _extra_tensors(self, sd):
        if "gradcam_weights" in sd:
            return sd
        sd = dict(sd)
        h = self.keras_input_size // 4  # two 2x2 pools before the final feature map (net/m7.py:41)
        affine = last_bn_affine(sd, self.root)
        alpha = grad_cam_alpha(sd["m7.classifier.0.weight"][:self.num_classes], h, h, "max",
                               bn_scale=None if affine is None else affine[0])
        sd["gradcam_weights"] = np.ascontiguousarray(alpha, dtype=np.float32)
        return sd
