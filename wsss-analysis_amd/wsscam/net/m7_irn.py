"""M7 IRNet -- mirror of 03b_irn/net/m7_irn.py (Net :8-118, EdgeDisplacement :195-213): three backbone stages,
the edge branch ends at 1/2 resolution (fc_edge4 on 3 x 32 channels), the displacement branch at 1/4; fc_dp4 is a
head on fc_dp3's output.  State-dict keys: `m7.layer<k>.*`, `fc_edge1..4.*`, `fc_dp1..5.*`,
`mean_shift.running_mean`."""
from .. import _lib
from .common_irn import EdgeDisplacementBase


class EdgeDisplacement(EdgeDisplacementBase):
    arch = _lib.ARCH_M7_IRN

    def __init__(self, model_dir=None, dataset="voc12", tag="", num_classes=20, use_cls=None, crop_size=512, stride=4,
                 precision=None):
        super().__init__(num_classes, crop_size, stride, precision)
        self.model_dir = model_dir
        self.dataset = dataset
        self.tag = tag
        self.use_cls = use_cls
