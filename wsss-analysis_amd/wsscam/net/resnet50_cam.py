"""ResNet50 CAM network -- mirror of 03b_irn/net/resnet50_cam.py.

Reference: Net.__init__ :12-20 (resnet50 with strides=(2,2,2,1) + 1x1 classifier), CAM.forward
:55-70.  The reference's CAM.__init__ takes no arguments and cannot be constructed the way
make_cam.run calls it; this class accepts make_cam's 5-argument call and the
bare call alike.  Weights come in through load_state_dict with the reference's keys
(`resnet50.*`, `classifier.weight`; the `stage*` / `backbone` / `newly_added` aliases of the
same tensors are ignored)."""
import numpy as np

from .. import _lib
from .common import DeviceCAMBase


class CAM(DeviceCAMBase):
    arch = _lib.ARCH_RESNET50_CAM

    def __init__(self, model_dir=None, dataset="voc12", tag="", num_classes=20, use_cls=None, precision=None):
        super().__init__(num_classes, precision)
        self.model_dir = model_dir
        self.dataset = dataset
        self.tag = tag
        self.use_cls = use_cls

    def forward(self, x):
        """x: (2,3,S,S) float32 [orig, h-flip] -> cam (C,h,w): relu(conv1x1) of x[0] + flipped x[1]."""
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        cam = self.forward_batch(xn[None])[0]
        if is_torch:
            import torch

            return torch.from_numpy(cam)
        return cam

    __call__ = forward
