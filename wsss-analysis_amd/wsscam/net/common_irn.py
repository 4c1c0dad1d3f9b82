"""Shared host logic of the IRNet EdgeDisplacement wrappers (03b_irn/net/*_irn.py, class EdgeDisplacement)."""
import numpy as np

from .. import _lib
from .common import DeviceCAMBase


class EdgeDisplacementBase(DeviceCAMBase):
    """EdgeDisplacement.forward (resnet50_irn.py:218-232, vgg16_irn.py:308-321): x (2,3,h,w) [orig, h-flip] is
    zero-padded on the right / bottom to crop_size, run through the backbone + edge / displacement heads,
    cropped to the stride-4 feature size, and returns (edge (1? no: (fh,fw) as edge_out[0]/2 + edge_out[1].flip(-1)/2
    has shape (1,fh,fw)), dp (2,fh,fw))."""

    def __init__(self, num_classes, crop_size=512, stride=4, precision=None):
        super().__init__(num_classes, precision)
        self.crop_size = crop_size
        self.stride = stride

    def forward_batch(self, x):
        """x: numpy/torch float32 (B,2,3,h,w) -> (edge (B,1,fh,fw), dp (B,2,fh,fw)) numpy."""
        net = self._ensure_net()
        xn = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        xn = np.ascontiguousarray(xn, dtype=np.float32)
        assert xn.ndim == 5 and xn.shape[1] == 2 and xn.shape[2] == 3, xn.shape
        B, h, w = xn.shape[0], xn.shape[3], xn.shape[4]
        S = self.crop_size
        if h > S or w > S:
            raise ValueError("input %dx%d is larger than crop_size %d" % (h, w, S))
        fh, fw = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        xp = np.zeros((B, 2, 3, S, S), np.float32)  # F.pad(x, [0, S - w, 0, S - h])
        xp[..., :h, :w] = xn
        ctx = self._ctx
        x_dev = ctx.to_device(xp)
        edge_dev = ctx.alloc(B * fh * fw * 4)
        dp_dev = ctx.alloc(B * 2 * fh * fw * 4)
        net.forward_edge(x_dev, B, S, fh, fw, edge_dev, dp_dev)
        edge = ctx.to_host(edge_dev, (B, 1, fh, fw), np.float32)
        dp = ctx.to_host(dp_dev, (B, 2, fh, fw), np.float32)
        return edge, dp

    def forward(self, x):
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        edge, dp = self.forward_batch(xn[None])
        if is_torch:
            import torch

            return torch.from_numpy(edge[0]), torch.from_numpy(dp[0])
        return edge[0], dp[0]

    __call__ = forward
