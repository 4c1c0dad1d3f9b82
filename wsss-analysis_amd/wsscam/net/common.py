"""Shared host logic of the CAM network wrappers (weights -> wsc_net, batched forward)."""
import numpy as np

from .. import _lib


def _to_numpy_sd(state_dict):
    out = {}
    for k, v in state_dict.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype.kind not in "fiu" or v.ndim > 4:
            continue
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class DeviceCAMBase:
    """Common plumbing: a lazily created wsc_ctx + wsc_net, numpy/torch in, same kind out."""

    arch = None
    precision = _lib.PREC_BF16

    def __init__(self, num_classes, precision=None):
        self.num_classes = num_classes
        if precision is not None:
            self.precision = precision
        self._sd = None
        self._ctx = None
        self._net = None
        self._device = 0
        self.training = False

    # -- nn.Module-like surface used by make_cam.run (make_cam.py:98-100, 33) ---------------
    def load_state_dict(self, state_dict, strict=True):
        self._sd = _to_numpy_sd(state_dict)
        self._release_net()
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self, device=None):
        if device is not None:
            self._device = int(device)
        self._ensure_net()
        return self

    def _release_net(self):
        if self._net is not None:
            self._net.close()
            self._net = None

    def _ensure_net(self):
        if self._net is None:
            if self._sd is None:
                raise RuntimeError("load_state_dict() must be called before the network is used")
            if self._ctx is None:
                self._ctx = _lib.Context(self._device)
            self._net = _lib.Net(self._ctx, self.arch, self._extra_tensors(self._sd), self.num_classes,
                                 self.precision)
        return self._net

    def _extra_tensors(self, sd):
        return sd

    @property
    def ctx(self):
        self._ensure_net()
        return self._ctx

    def cam_size(self, S):
        return self._ensure_net().cam_size(S)

    def gradcam_net(self, weights):
        """(wsc_net with `weights` (F x C Grad-CAM alpha) as its 1x1 head, ctx); cached per alpha."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        key = (w.shape, hash(w.tobytes()))
        cache = self.__dict__.setdefault("_gradcam_nets", {})
        if key not in cache:
            if self._ctx is None:
                self._ctx = _lib.Context(self._device)
            sd = dict(self._sd)
            sd["gradcam_weights"] = w
            cache[key] = _lib.Net(self._ctx, self.arch, sd, w.shape[1], self.precision)
        return cache[key], self._ctx

    # -- batched device forward ---------------------------------------------------------------
    def forward_batch_device(self, x_dev, B, S, cam_dev, score_dev=None):
        """x_dev float32 [B][2][3][S][S] -> cam_dev float32 [B][C][h][w] (device pointers/buffers)."""
        self._ensure_net().forward_cam(x_dev, B, S, cam_dev, score_dev)

    def forward_batch(self, x, want_score=False):
        """x: numpy/torch float32 (B,2,3,S,S) on host -> numpy cam (B,C,h,w) [, score (B,C)]."""
        net = self._ensure_net()
        is_torch = hasattr(x, "detach")
        xn = x.detach().cpu().numpy() if is_torch else np.asarray(x)
        xn = np.ascontiguousarray(xn, dtype=np.float32)
        assert xn.ndim == 5 and xn.shape[1] == 2 and xn.shape[2] == 3 and xn.shape[3] == xn.shape[4], xn.shape
        B, S = xn.shape[0], xn.shape[3]
        h = net.cam_size(S)
        ctx = self._ctx
        x_dev = ctx.to_device(xn)
        cam_dev = ctx.alloc(B * self.num_classes * h * h * 4)
        score_dev = ctx.alloc(B * self.num_classes * 4) if want_score else None
        net.forward_cam(x_dev, B, S, cam_dev, score_dev)
        cam = ctx.to_host(cam_dev, (B, self.num_classes, h, h), np.float32)
        score = ctx.to_host(score_dev, (B, self.num_classes), np.float32) if want_score else None
        return (cam, score) if want_score else cam
