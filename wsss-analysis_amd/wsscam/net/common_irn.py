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
        edge_dev, dp_dev, (B, fh, fw) = self.forward_batch_device(x)
        ctx = self._ctx
        return ctx.to_host(edge_dev, (B, 1, fh, fw), np.float32), ctx.to_host(dp_dev, (B, 2, fh, fw), np.float32)

    def forward_batch_device(self, x, ctx=None, chain=None):
        """forward_batch with the outputs left in HBM: (edge_dev float32 [B][fh][fw], dp_dev [B][2][fh][fw], (B, fh, fw)).
        x: (B,2,3,h,w) array / tensor, or a list of B (2,3,h,w) arrays of one size (copied straight into the page-locked,
        zero-padded staging batch: no stacked temporary, no pageable upload).  `ctx`: run on this context (a lane of the
        make_sem_seg_labels driver: its own stream, staging batch and workspace) instead of the model's own; `chain`: that
        driver's _lib.StackChain (the lanes' network passes take turns on the device)."""
        net = self._ensure_net()
        if isinstance(x, (list, tuple)):
            items = [np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32) for v in x]
        else:
            xn = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
            assert xn.ndim == 5, xn.shape
            items = list(np.asarray(xn, dtype=np.float32))
        B = len(items)
        assert B > 0 and all(v.shape == items[0].shape and v.ndim == 4 and v.shape[:2] == (2, 3) for v in items), \
            [v.shape for v in items][:4]
        h, w = items[0].shape[2], items[0].shape[3]
        S = self.crop_size
        if h > S or w > S:
            raise ValueError("input %dx%d is larger than crop_size %d" % (h, w, S))
        fh, fw = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        ctx = ctx or self._ctx
        st = self.__dict__.setdefault("_stage", {}).setdefault(id(ctx), {})
        if st.get("key") != (B, S):  # page-locked (B,2,3,S,S) staging batch + its device twin, kept across calls
            for k in ("pin", "dev"):
                if st.get(k) is not None:
                    st[k].free()
            st.update(key=(B, S), pin=ctx.host_alloc(B * 2 * 3 * S * S * 4), dev=ctx.alloc(B * 2 * 3 * S * S * 4), hw=None)
        xp = st["pin"].view((B, 2, 3, S, S), np.float32)
        ctx.sync()  # the previous batch's upload has left the staging buffer
        if st["hw"] != (h, w):
            xp[...] = 0.0  # F.pad(x, [0, S - w, 0, S - h]): the margins stay zero while the image size does not change
            st["hw"] = (h, w)
        for i, v in enumerate(items):
            xp[i, :, :, :h, :w] = v
        ctx.h2d_async(st["dev"], st["pin"], B * 2 * 3 * S * S * 4)
        edge_dev = ctx.alloc(B * fh * fw * 4, pooled=True)
        dp_dev = ctx.alloc(B * 2 * fh * fw * 4, pooled=True)
        if chain is not None:  # (a driver with several batches in flight: one conv stack at a time, _lib.StackChain)
            chain.run(ctx, lambda: net.forward_edge(st["dev"], B, S, fh, fw, edge_dev, dp_dev, ctx=ctx))
        else:
            net.forward_edge(st["dev"], B, S, fh, fw, edge_dev, dp_dev, ctx=ctx)
        return edge_dev, dp_dev, (B, fh, fw)

    def forward(self, x):
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        edge, dp = self.forward_batch(xn[None])
        if is_torch:
            import torch

            return torch.from_numpy(edge[0]), torch.from_numpy(dp[0])
        return edge[0], dp[0]

    __call__ = forward
