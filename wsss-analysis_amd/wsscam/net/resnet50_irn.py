"""ResNet50 IRNet -- mirror of 03b_irn/net/resnet50_irn.py (Net :8-132, EdgeDisplacement :210-232).
State-dict keys: `resnet50.*` backbone (strides [2,2,2,1]), `fc_edge<k>.*`, `fc_dp<k>.*`,
`mean_shift.running_mean` (the `stage*`, `backbone.*`, `edge_layers.*`, `dp_layers.*` aliases of the same
tensors that nn.Module.state_dict() also emits are ignored)."""
from .. import _lib
from .common_irn import EdgeDisplacementBase


class EdgeDisplacement(EdgeDisplacementBase):
    arch = _lib.ARCH_RESNET50_IRN

    def __init__(self, model_dir=None, num_classes=20, crop_size=512, stride=4, precision=None):
        super().__init__(num_classes, crop_size, stride, precision)
        self.model_dir = model_dir
