"""M7 Grad-CAM network -- mirror of 03b_irn/net/m7_cam.py (CAM.forward :22-57) on net/m7.py:41.
The 1x1 head is the Grad-CAM weight matrix (F x C) computed once per model on a zeros image
(common_cnn.gen_gradcam_weights :84-121); pass it in the state_dict as `gradcam_weights`."""
from .. import _lib
from . import vgg16_cam


class CAM(vgg16_cam.CAM):
    arch = _lib.ARCH_M7_CAM
    root = "m7"
