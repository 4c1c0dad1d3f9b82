"""M7 Grad-CAM network -- mirror of 03b_irn/net/m7_cam.py (CAM.forward :22-57) on net/m7.py:41.
The 1x1 head is the Grad-CAM weight matrix (F x C) computed once per model on a zeros image
(common_cnn.gen_gradcam_weights :84-121, run by MNet.__init__ through `_load_pretrained(gen_gradcam=True)`,
net/m7.py:22).  A state dict without `gradcam_weights` gets them here the same way -- from the closed form of
net.common.grad_cam_alpha on the Keras model's 224 x 224 input (56 x 56 x 256 final feature map)."""
import numpy as np

from .. import _lib
from . import vgg16_cam
from .common import grad_cam_alpha


class CAM(vgg16_cam.CAM):
    arch = _lib.ARCH_M7_CAM
    root = "m7"
    keras_input_size = 224  # 02_cues/demo.py:60-66, 03b_irn/func_sample.py:151-156

    def _extra_tensors(self, sd):
        if "gradcam_weights" in sd:
            return sd
        sd = dict(sd)
        h = self.keras_input_size // 4  # two 2x2 pools before the final feature map (net/m7.py:41)
        alpha = grad_cam_alpha(sd["m7.classifier.0.weight"][:self.num_classes], h, h, "max")
        sd["gradcam_weights"] = np.ascontiguousarray(alpha, dtype=np.float32)
        return sd
