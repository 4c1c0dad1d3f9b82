"""Modified-VGG16 IRNet -- mirror of 03b_irn/net/vgg16_irn.py (Net :8-212 with ds_fac = 0.25,
EdgeDisplacement :301-321).  State-dict keys: `vgg16.layer<k>.*` backbone (BatchNorm absent for the ADP
datasets), `fc_edge<k>.*`, `fc_dp<k>.*`, `mean_shift.running_mean`."""
from .. import _lib
from .common_irn import EdgeDisplacementBase


class EdgeDisplacement(EdgeDisplacementBase):
    arch = _lib.ARCH_VGG16_IRN

    def __init__(self, model_dir=None, dataset="voc12", tag="", num_classes=20, use_cls=None, crop_size=512, stride=4,
                 precision=None):
        super().__init__(num_classes, crop_size, stride, precision)
        self.model_dir = model_dir
        self.dataset = dataset
        self.tag = tag
        self.use_cls = use_cls
