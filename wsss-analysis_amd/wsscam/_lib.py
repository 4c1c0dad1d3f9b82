"""ctypes binding of libwsscam.so (the C ABI of include/wsscam.h).

The product path has no CPU fallback: if the shared library is missing, or no gfx950
device is present, the calls fail loudly (WscError).
"""
import ctypes
import os
import re
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwsscam.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "wsscam.h"))

WSC_OK = 0
WSC_ERR_INVALID = -1
WSC_ERR_NO_DEVICE = -2
WSC_ERR_HIP = -3
WSC_ERR_MISSING_KEY = -4
WSC_ERR_SHAPE = -5
WSC_ERR_NOMEM = -6
WSC_ERR_KEY_RANGE = -7
WSC_ERR_CAPACITY = -8
WSC_ERR_RANGE = -9  # an IEEE-half activation saturated (|v| >= 65504): the maps are not the reference's fp32 maps

ARCH_RESNET50_CAM = 0
ARCH_VGG16_CAM = 1
ARCH_M7_CAM = 2
ARCH_RESNET50_IRN = 3
ARCH_VGG16_IRN = 4
ARCH_M7_IRN = 5

PREC_BF16 = 0
PREC_BF16X3 = 1
PREC_F16 = 2
PREC_F16X3 = 3  # split-half operands and activations (hi + lo), three MFMA products: the fp32-class mode
# path selectors of a context (include/wsscam.h wsc_option; Context.set_option / Context.option)
OPT_CRF_GAUSS_ON_CHIP, OPT_CRF_FUSED_BLUR, OPT_CRF_BLUR_ON_CHIP, OPT_CRF_RANK_BALLOT, OPT_CRF_EMBED_FULL, OPT_RW_TILED, \
    OPT_STEM_POOL_FUSED, OPT_CONV_WINDOW, OPT_CAM_HEAD_STREAM, OPT_CRF_MSG_IN_UPDATE = range(10)
OPT_DEFAULTS = {OPT_CRF_GAUSS_ON_CHIP: 1, OPT_CRF_FUSED_BLUR: 1, OPT_CRF_BLUR_ON_CHIP: 1, OPT_CRF_RANK_BALLOT: 0,
                OPT_CRF_EMBED_FULL: 0, OPT_RW_TILED: -1, OPT_STEM_POOL_FUSED: 1, OPT_CONV_WINDOW: 1, OPT_CAM_HEAD_STREAM: 1,
                OPT_CRF_MSG_IN_UPDATE: 1}
CONV_GENERIC = 0x100  # conv2d_nchw only: OR into precision to keep the kernel's generic variants (testing)


class WscError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libwsscam error %d: %s" % (status, message))
        self.status = status


class TensorDesc(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char_p),
        ("data", ctypes.POINTER(ctypes.c_float)),
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int64 * 4),
    ]


_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/wsscam.h declares
_SIGNATURES = {
    "wsc_version": (_i, []),
    "wsc_last_error": (ctypes.c_char_p, []),
    "wsc_ctx_create": (_i, [_i, _vp, ctypes.POINTER(_vp)]),
    "wsc_ctx_destroy": (None, [_vp]),
    "wsc_sync": (_i, [_vp]),
    "wsc_ctx_range_status": (_i, [_vp, ctypes.POINTER(_i), _i]),
    "wsc_ctx_wait": (_i, [_vp, _vp]),
    "wsc_ctx_mark": (_i, [_vp, _i]),
    "wsc_ctx_wait_mark": (_i, [_vp, _i]),
    "wsc_ctx_wait_for_mark": (_i, [_vp, _vp, _i]),
    "wsc_device_info": (_i, [_vp, ctypes.c_char_p, _sz, ctypes.POINTER(_i)]),
    "wsc_malloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "wsc_free": (_i, [_vp, _vp]),
    "wsc_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "wsc_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "wsc_memset": (_i, [_vp, _vp, _i, _sz]),
    "wsc_ctx_set_option": (_i, [_vp, _i, _i]),
    "wsc_host_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "wsc_host_free": (_i, [_vp, _vp]),
    "wsc_host_write_segments": (_i, [ctypes.c_char_p, _i, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]),
    "wsc_memcpy_h2d_async": (_i, [_vp, _vp, _vp, _sz]),
    "wsc_memcpy_d2h_async": (_i, [_vp, _vp, _vp, _sz]),
    "wsc_profile_begin": (_i, [_vp]),
    "wsc_profile_end": (_i, [_vp, _i, _vp, _vp, _vp, ctypes.POINTER(_i)]),
    "wsc_profile_class_name": (ctypes.c_char_p, [_i]),
    "wsc_timer_begin": (_i, [_vp]),
    "wsc_timer_end": (_i, [_vp, ctypes.POINTER(_f)]),
    "wsc_net_create": (_i, [_vp, _i, ctypes.POINTER(TensorDesc), _i, _i, _i, ctypes.POINTER(_vp)]),
    "wsc_net_destroy": (None, [_vp]),
    "wsc_net_cam_size": (_i, [_vp, _i, ctypes.POINTER(_i)]),
    "wsc_net_feat_channels": (_i, [_vp, ctypes.POINTER(_i)]),
    "wsc_net_forward_cam": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "wsc_net_forward_gradcam": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "wsc_net_forward_features": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "wsc_net_cam_size_hw": (_i, [_vp, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "wsc_net_forward_cam_hw": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "wsc_net_forward_edge": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "wsc_rw_propagate": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "wsc_rw_propagate_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "wsc_conv2d_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "wsc_cam_postprocess": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wsc_cam_eval_confusion": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _vp, _vp]),
    "wsc_unary_from_maps": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "wsc_cam_unary": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "wsc_cam_unary_pm": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "wsc_cam_sum_scales": (_i, [_vp, _vp, _i, _i, ctypes.c_longlong, _vp]),
    "wsc_cam_eval_confusion_nn": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "wsc_label_confusion_nn": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "wsc_sem_seg_finish": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "wsc_bilinear_resize": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i]),
    "wsc_msf_input_u8": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "wsc_resize_u8": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "wsc_label_unary_from_cam": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    "wsc_ir_label_combine": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "wsc_hsn_gradcam_post": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i]),
    "wsc_hsn_voc_background": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "wsc_hsn_class_mass": (_i, [_vp, _vp, _i, _i, _vp]),
    "wsc_hsn_background": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "wsc_cam_adp_modify": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "wsc_hsn_cs_gradcam": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "wsc_hsn_gather_unary": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "wsc_crf_create": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _f, ctypes.POINTER(_vp)]),
    "wsc_crf_destroy": (None, [_vp]),
    "wsc_crf_lattice_sizes": (_i, [_vp, _vp, _vp, _vp]),
    "wsc_crf_gaussian_on_chip": (_i, [_vp, _i]),
    "wsc_crf_inference": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp]),
    "wsc_crf_inference_pm": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp]),
    "wsc_crf_v_create": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, ctypes.POINTER(_vp)]),
    "wsc_crf_v_destroy": (None, [_vp]),
    "wsc_crf_v_num_groups": (_i, [_vp]),
    "wsc_crf_v_inference": (_i, [_vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp]),
}


def header_symbols():
    """Function names declared in include/wsscam.h."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wsc_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libwsscam.so (no GPU needed for loading; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WscError(WSC_ERR_NO_DEVICE,
                       "libwsscam.so not built at %s -- run `python __graft_entry__.py` (there is no "
                       "CPU fallback for the product path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check_exports():
    """Every symbol the header declares must be exported and bound."""
    lib = load()
    declared = header_symbols()
    missing = [s for s in declared if not hasattr(lib, s)]
    unbound = [s for s in declared if s not in _SIGNATURES]
    if missing or unbound:
        raise WscError(WSC_ERR_INVALID, "missing exports %s / unbound %s" % (missing, unbound))
    return declared


def check(status):
    if status != WSC_OK:
        raise WscError(status, load().wsc_last_error().decode("utf-8", "replace"))


def _ptr(x):
    """Device pointer of a DeviceBuffer / torch tensor / int, or host pointer of a numpy array."""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):  # torch tensor
        return x.data_ptr()
    raise TypeError("cannot take a pointer of %r" % type(x))


class StackChain:
    """One GPU-filling phase at a time across the contexts of a driver's lanes.  `run(ctx, enqueue)` makes what `enqueue()` puts
    on ctx's stream wait -- on the device, wsc_ctx_wait_for_mark -- for the phase the chain ran last on another context.  A conv
    stack fills the GPU by itself: two of them interleaved by the hardware scheduler ran 8.5 ms each = 7.1 ms per batch where
    back to back they take 6.3 (step.make_cam through step.pipeline, round 6: 4490 -> 5110 images/s); the other lanes' uploads,
    tails, CRFs and read-backs still overlap the running stack.  Marker slot 7 of the lanes' contexts belongs to the chain."""

    SLOT = 7

    def __init__(self):
        self._lock = threading.Lock()
        self._last = None

    def run(self, ctx, enqueue):
        with self._lock:
            prev = self._last
            if prev is not None and prev is not ctx:
                ctx.wait_for_mark(prev, self.SLOT)
            try:
                return enqueue()
            finally:
                ctx.mark(self.SLOT)
                self._last = ctx


class Context:
    """wsc_ctx: one device + one stream (make_cam.py:31-33 `cuda.device(process_id)`)."""

    def __init__(self, device=0, stream=None):
        lib = load()
        h = _vp()
        check(lib.wsc_ctx_create(int(device), stream, ctypes.byref(h)))
        self.h = h
        self.device = device
        self._lib = lib
        self._options = {}  # selectors set through set_option (the library has no getter; option() restores from here)

    def close(self):
        if getattr(self, "h", None):
            pool = self.__dict__.pop("_pool", None)
            if pool:
                for lst in pool.values():
                    for buf in lst:
                        self._lib.wsc_free(self.h, buf.ptr)
                        buf.ptr = None
            self._lib.wsc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.wsc_sync(self.h))

    def range_status(self, clear=False):
        """Range guard of the IEEE-half conv modes: 0 = clean, else the output-channel count of a layer whose activations
        reached half's ceiling and were saturated (sync() / to_host() raise WscError(WSC_ERR_RANGE) until cleared)."""
        f = _i(0)
        check(self._lib.wsc_ctx_range_status(self.h, ctypes.byref(f), 1 if clear else 0))
        return int(f.value)

    def mark(self, slot):
        """Records marker `slot` (0 .. 7) at the current end of this context's stream (wsc_ctx_mark)."""
        check(self._lib.wsc_ctx_mark(self.h, int(slot)))

    def wait_mark(self, slot):
        """Host wait for everything enqueued before the marker's last record (returns at once if never recorded)."""
        check(self._lib.wsc_ctx_wait_mark(self.h, int(slot)))

    def wait_for_mark(self, other, slot):
        """Device-side join with marker `slot` of `other`: later work on this ctx waits for what `other` had enqueued when it
        last recorded the marker -- not for anything it enqueued afterwards."""
        check(self._lib.wsc_ctx_wait_for_mark(self.h, other.h, int(slot)))

    def wait_for(self, other):
        """Device-side join: later work on this ctx waits for everything enqueued so far on `other`."""
        check(self._lib.wsc_ctx_wait(self.h, other.h))

    def device_info(self):
        buf = ctypes.create_string_buffer(128)
        n = _i()
        check(self._lib.wsc_device_info(self.h, buf, 128, ctypes.byref(n)))
        return buf.value.decode(), n.value

    def alloc(self, nbytes, pooled=False):
        """pooled=True: the block comes from / returns to a per-context free list instead of hipMalloc / hipFree (both
        synchronise the device).  Only for buffers that are used on THIS context's stream alone: reuse is ordered by
        the stream, exactly like the library's internal cached allocator."""
        nbytes = int(nbytes)
        if pooled:
            size = 1 << max(16, (max(nbytes, 1) - 1).bit_length())
            pool = self.__dict__.setdefault("_pool", {})
            lst = pool.get(size)
            if lst:
                buf = lst.pop()
                buf.nbytes = nbytes
                return buf
            buf = DeviceBuffer(self, size)
            buf.nbytes, buf._pool_size = nbytes, size
            return buf
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr, pooled=False):
        arr = np.ascontiguousarray(arr)
        buf = self.alloc(arr.nbytes, pooled=pooled)
        check(self._lib.wsc_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes))
        self.sync()  # arr may be a temporary
        return buf

    def to_host(self, buf, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        check(self._lib.wsc_memcpy_d2h(self.h, out.ctypes.data, _ptr(buf) + offset_bytes, out.nbytes))
        return out

    def host_alloc(self, nbytes):
        """Page-locked host buffer (wsc_host_alloc) for asynchronous copies."""
        return PinnedBuffer(self, int(nbytes))

    def h2d_async(self, dst_dev, pinned, nbytes, dst_offset=0, src_offset=0):
        check(self._lib.wsc_memcpy_h2d_async(self.h, _ptr(dst_dev) + dst_offset, pinned.ptr + src_offset, int(nbytes)))

    def d2h_async(self, pinned, src_dev, nbytes, dst_offset=0, src_offset=0):
        check(self._lib.wsc_memcpy_d2h_async(self.h, pinned.ptr + dst_offset, _ptr(src_dev) + src_offset, int(nbytes)))

    def set_option(self, option, value):
        """A path selector of this context (OPT_*): forces a fallback path that gives the same bits (testing / debugging) --
        except OPT_CAM_HEAD_STREAM, whose two paths sum K in different orders and agree to fp32 round-off (2e-6 relative) only."""
        check(self._lib.wsc_ctx_set_option(self.h, int(option), int(value)))
        self._options[int(option)] = int(value)

    def get_option(self, option):
        """The selector's current value (what set_option last stored; OPT_DEFAULTS before that)."""
        return self._options.get(int(option), OPT_DEFAULTS[int(option)])

    def option(self, option, value):
        """Context manager: the selector is set inside the block and back to the value it HAD afterwards (an earlier
        set_option or an enclosing option() block stays in force)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            before = self.get_option(option)
            self.set_option(option, value)
            try:
                yield self
            finally:
                self.set_option(option, before)

        return _cm()

    def profile_begin(self):
        check(self._lib.wsc_profile_begin(self.h))

    def profile_end(self):
        """-> {class name: (launches, total_ms, algorithmic work)} for the launches since profile_begin."""
        n = 16
        calls = np.zeros(n, np.int32)
        ms = np.zeros(n, np.float32)
        work = np.zeros(n, np.float64)
        k = _i()
        check(self._lib.wsc_profile_end(self.h, n, calls.ctypes.data, ms.ctypes.data, work.ctypes.data, ctypes.byref(k)))
        return {self._lib.wsc_profile_class_name(i).decode(): (int(calls[i]), float(ms[i]), float(work[i]))
                for i in range(k.value) if calls[i] > 0}

    def timer_begin(self):
        check(self._lib.wsc_timer_begin(self.h))

    def timer_end(self):
        ms = _f()
        check(self._lib.wsc_timer_end(self.h, ctypes.byref(ms)))
        return ms.value


class DeviceBuffer:
    """hipMalloc'ed bytes owned through wsc_malloc / wsc_free."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = nbytes
        p = _vp()
        check(ctx._lib.wsc_malloc(ctx.h, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            size = getattr(self, "_pool_size", None)
            if size is not None and getattr(self.ctx, "_pool", None) is not None:
                twin = DeviceBuffer.__new__(DeviceBuffer)  # the block lives on in the pool under a fresh handle
                twin.ctx, twin.nbytes, twin.ptr, twin._pool_size = self.ctx, size, self.ptr, size
                self.ctx._pool.setdefault(size, []).append(twin)
                self.ptr = None
                return
            self.ctx._lib.wsc_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """hipHostMalloc'ed bytes; `.view(shape, dtype, offset)` is a numpy array over them (no copy)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = nbytes
        p = _vp()
        check(ctx._lib.wsc_host_alloc(ctx.h, nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self._raw = (ctypes.c_char * max(nbytes, 1)).from_address(self.ptr)

    def view(self, shape, dtype, offset_bytes=0):
        n = int(np.prod(shape)) if len(shape) else 1
        return np.frombuffer(self._raw, dtype=dtype, count=n, offset=offset_bytes).reshape(shape)

    def free(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            self._raw = None
            self.ctx._lib.wsc_host_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_write_segments(path, segments):
    """wsc_host_write_segments: `segments` = bytes objects / C-contiguous numpy arrays, written to `path` one after the other by
    one C call (open + writev + close) -- the interpreter lock is released for its whole duration."""
    n = len(segments)
    ptrs = (_vp * max(n, 1))()
    sizes = (_sz * max(n, 1))()
    keep = []
    for i, sgm in enumerate(segments):
        if isinstance(sgm, np.ndarray):
            assert sgm.flags["C_CONTIGUOUS"]
            ptrs[i], sizes[i] = sgm.ctypes.data, sgm.nbytes
        else:
            sgm = bytes(sgm)
            keep.append(sgm)  # (the pointer below is the bytes object's own buffer)
            ptrs[i], sizes[i] = ctypes.cast(ctypes.c_char_p(sgm), _vp).value, len(sgm)
    check(load().wsc_host_write_segments(os.fsencode(path), n, ptrs, sizes))


def make_tensor_descs(state_dict):
    """dict name -> numpy float32 array  ->  (ctypes array of TensorDesc, keep-alive list)."""
    keep = []
    items = []
    for name, arr in state_dict.items():
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        if a.ndim > 4:
            continue
        keep.append(a)
        d = TensorDesc()
        d.name = name.encode()
        d.data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        d.ndim = a.ndim
        for k in range(4):
            d.shape[k] = a.shape[k] if k < a.ndim else 1
        items.append(d)
    arr_t = TensorDesc * len(items)
    return arr_t(*items), keep


class Net:
    """wsc_net: packed weights of one CAM network."""

    def __init__(self, ctx, arch, state_dict, num_classes, precision=PREC_F16X3):
        self.ctx = ctx
        self.arch = arch
        self.num_classes = num_classes
        self.precision = precision
        descs, keep = make_tensor_descs(state_dict)
        h = _vp()
        check(ctx._lib.wsc_net_create(ctx.h, arch, descs, len(descs), num_classes, precision, ctypes.byref(h)))
        del keep
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.sync()
            self.ctx._lib.wsc_net_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cam_size(self, S):
        n = _i()
        check(self.ctx._lib.wsc_net_cam_size(self.h, S, ctypes.byref(n)))
        return n.value

    def feat_channels(self):
        n = _i()
        check(self.ctx._lib.wsc_net_feat_channels(self.h, ctypes.byref(n)))
        return n.value

    def forward_cam(self, x_dev, B, S, cam_dev, score_dev=None, ctx=None):
        """ctx: the context (stream, activation arena) to run on; the packed weights are read-only and shared."""
        run = ctx or self.ctx
        check(self.ctx._lib.wsc_net_forward_cam(run.h, self.h, _ptr(x_dev), B, S, _ptr(cam_dev),
                                                _ptr(score_dev)))

    def cam_size_hw(self, H, W):
        h, w = _i(), _i()
        check(self.ctx._lib.wsc_net_cam_size_hw(self.h, int(H), int(W), ctypes.byref(h), ctypes.byref(w)))
        return h.value, w.value

    def forward_cam_hw(self, x_dev, B, H, W, cam_dev, score_dev=None, ctx=None):
        """Non-square network input [B][2][3][H][W] (outsize = None: the image's own size)."""
        run = ctx or self.ctx
        check(self.ctx._lib.wsc_net_forward_cam_hw(run.h, self.h, _ptr(x_dev), B, int(H), int(W), _ptr(cam_dev), _ptr(score_dev)))

    def forward_gradcam(self, x_dev, N, S, relu, cams_dev, score_dev=None, ctx=None):
        run = ctx or self.ctx
        check(self.ctx._lib.wsc_net_forward_gradcam(run.h, self.h, _ptr(x_dev), N, S, int(relu), _ptr(cams_dev),
                                                    _ptr(score_dev)))

    def forward_features(self, x_dev, N, S, feat_dev):
        check(self.ctx._lib.wsc_net_forward_features(self.ctx.h, self.h, _ptr(x_dev), N, S, _ptr(feat_dev)))

    def forward_edge(self, x_dev, B, S, feat_h, feat_w, edge_dev, dp_dev, ctx=None):
        run = ctx or self.ctx
        check(self.ctx._lib.wsc_net_forward_edge(run.h, self.h, _ptr(x_dev), B, S, feat_h, feat_w,
                                                 _ptr(edge_dev), _ptr(dp_dev)))


def cam_postprocess(ctx, cam_dev, B, C, h, w, sizes, keys_per_image, strided_dev=None, highres_dev=None):
    """Batched make_cam tail.  sizes: [(H0, W0)], keys_per_image: list of int sequences.

    Returns (strided_dev, highres_dev, strided_off, highres_off, shapes) where shapes[b] =
    (K, h4, w4, H0, W0); buffers are allocated when not given.
    """
    size_hw = np.asarray(sizes, dtype=np.int32).reshape(B, 2)
    key_off = np.zeros(B + 1, dtype=np.int32)
    for b in range(B):
        key_off[b + 1] = key_off[b] + len(keys_per_image[b])
    keys = np.zeros(max(int(key_off[-1]), 1), dtype=np.int32)
    for b in range(B):
        keys[key_off[b]:key_off[b + 1]] = np.asarray(keys_per_image[b], dtype=np.int32)
    s_off = np.zeros(B, dtype=np.int64)
    h_off = np.zeros(B, dtype=np.int64)
    shapes = []
    s_tot = 0
    h_tot = 0
    for b in range(B):
        H0, W0 = int(size_hw[b, 0]), int(size_hw[b, 1])
        h4, w4 = (H0 - 1) // 4 + 1, (W0 - 1) // 4 + 1
        K = int(key_off[b + 1] - key_off[b])
        s_off[b] = s_tot
        h_off[b] = h_tot
        s_tot += K * h4 * w4
        h_tot += K * H0 * W0
        shapes.append((K, h4, w4, H0, W0))
    if strided_dev is None:
        strided_dev = ctx.alloc(max(s_tot, 1) * 4)
    if highres_dev is None:
        highres_dev = ctx.alloc(max(h_tot, 1) * 4)
    check(ctx._lib.wsc_cam_postprocess(ctx.h, _ptr(cam_dev), B, C, h, w, size_hw.ctypes.data, keys.ctypes.data,
                                       key_off.ctypes.data, s_off.ctypes.data, h_off.ctypes.data,
                                       _ptr(strided_dev), _ptr(highres_dev)))
    return strided_dev, highres_dev, s_off, h_off, shapes


def conv2d_nchw(ctx, x_dev, N, Cin, H, W, w, stride, pad, scale=None, shift=None, residual_dev=None, relu=False,
                precision=PREC_BF16, y_dev=None):
    """One conv layer through the production kernel (test/diagnostic entry)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    Cout, _, kh, kw = w.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if y_dev is None:
        y_dev = ctx.alloc(N * Cout * Ho * Wo * 4)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    check(ctx._lib.wsc_conv2d_nchw(ctx.h, _ptr(x_dev), N, Cin, H, W, w.ctypes.data, Cout, kh, kw, stride, pad,
                                   None if sc is None else sc.ctypes.data, None if sh is None else sh.ctypes.data,
                                   _ptr(residual_dev), int(relu), precision, _ptr(y_dev)))
    return y_dev, (N, Cout, Ho, Wo)


def cam_eval_confusion(ctx, highres_dev, sizes, keys_per_image, highres_off, bg_thres, gt_dev, n_class, confusion_dev,
                       pred_dev=None, ignore_label=255):
    """Accumulates the eval_cam confusion matrix of a batch into confusion_dev (int64 [n_class][n_class])."""
    B = len(sizes)
    size_hw = np.asarray(sizes, dtype=np.int32).reshape(B, 2)
    key_off = np.zeros(B + 1, dtype=np.int32)
    for b in range(B):
        key_off[b + 1] = key_off[b] + len(keys_per_image[b])
    keys = np.zeros(max(int(key_off[-1]), 1), dtype=np.int32)
    for b in range(B):
        keys[key_off[b]:key_off[b + 1]] = np.asarray(keys_per_image[b], dtype=np.int32)
    h_off = np.ascontiguousarray(highres_off, dtype=np.int64)
    check(ctx._lib.wsc_cam_eval_confusion(ctx.h, _ptr(highres_dev), B, size_hw.ctypes.data, keys.ctypes.data,
                                          key_off.ctypes.data, h_off.ctypes.data, float(bg_thres), _ptr(gt_dev),
                                          int(n_class), int(ignore_label), _ptr(pred_dev), _ptr(confusion_dev)))


def cam_eval_confusion_nn(ctx, maps_dev, src_sizes, out_sizes, keys_per_image, maps_off, gt_dev, n_class, confusion_dev,
                          pred_dev=None, ignore_label=255):
    """ADP / DeepGlobe eval_cam branch: keys[argmax(maps)] at the maps' size, cv2 nearest resize to out_sizes, confusion."""
    B = len(src_sizes)
    src_hw = np.asarray(src_sizes, dtype=np.int32).reshape(B, 2)
    out_hw = np.asarray(out_sizes, dtype=np.int32).reshape(B, 2)
    key_off = np.zeros(B + 1, dtype=np.int32)
    for b in range(B):
        key_off[b + 1] = key_off[b] + len(keys_per_image[b])
    keys = np.zeros(max(int(key_off[-1]), 1), dtype=np.int32)
    for b in range(B):
        keys[key_off[b]:key_off[b + 1]] = np.asarray(keys_per_image[b], dtype=np.int32)
    m_off = np.ascontiguousarray(maps_off, dtype=np.int64)
    check(ctx._lib.wsc_cam_eval_confusion_nn(ctx.h, _ptr(maps_dev), B, src_hw.ctypes.data, out_hw.ctypes.data, keys.ctypes.data,
                                             key_off.ctypes.data, m_off.ctypes.data, _ptr(gt_dev), int(n_class),
                                             int(ignore_label), _ptr(pred_dev), _ptr(confusion_dev)))


def label_confusion_nn(ctx, labels_dev, src_sizes, out_sizes, labels_off, gt_dev, n_class, confusion_dev, pred_dev=None,
                       ignore_label=255):
    """HSN evaluation tail: int32 label maps -> cv2 nearest resize to out_sizes -> confusion[gt][pred] (accumulated)."""
    B = len(src_sizes)
    src_hw = np.asarray(src_sizes, dtype=np.int32).reshape(B, 2)
    out_hw = np.asarray(out_sizes, dtype=np.int32).reshape(B, 2)
    off = np.ascontiguousarray(labels_off, dtype=np.int64)
    check(ctx._lib.wsc_label_confusion_nn(ctx.h, _ptr(labels_dev), B, src_hw.ctypes.data, out_hw.ctypes.data, off.ctypes.data,
                                          _ptr(gt_dev), int(n_class), int(ignore_label), _ptr(pred_dev), _ptr(confusion_dev)))


def sem_seg_finish(ctx, rw_dev, rw_off, khw, up_hw, out_hw, keys_per_image, has_bg, bg_thres, label_dev):
    """make_sem_seg_labels tail on the device: upsample + crop + / max + [bg pad] + argmax + keys -> packed uint8 labels."""
    B = len(khw)
    key_off = np.zeros(B + 1, dtype=np.int32)
    for b in range(B):
        key_off[b + 1] = key_off[b] + len(keys_per_image[b])
    keys = np.zeros(max(int(key_off[-1]), 1), dtype=np.int32)
    for b in range(B):
        keys[key_off[b]:key_off[b + 1]] = np.asarray(keys_per_image[b], dtype=np.int32)
    off = np.ascontiguousarray(rw_off, dtype=np.int64)
    a, u, o = (np.ascontiguousarray(v, dtype=np.int32).reshape(B, -1) for v in (khw, up_hw, out_hw))
    check(ctx._lib.wsc_sem_seg_finish(ctx.h, _ptr(rw_dev), B, off.ctypes.data, a.ctypes.data, u.ctypes.data, o.ctypes.data,
                                      keys.ctypes.data, key_off.ctypes.data, int(bool(has_bg)), float(bg_thres), _ptr(label_dev)))


def unary_from_maps(ctx, maps_dev, B, C, N, bg_value, unary_dev):
    check(ctx._lib.wsc_unary_from_maps(ctx.h, _ptr(maps_dev), B, C, N, float(bg_value), _ptr(unary_dev)))


def cam_unary(ctx, cam_dev, B, C, h, w, H0, W0, bg_value, unary_dev, pixel_major=False):
    """cam_postprocess (all classes at H0 x W0) + unary_from_maps without the intermediate maps in HBM.
    pixel_major: unaries as [B][H0*W0][Mp] (Mp = 4 ceil((C+1)/4)) for Crf.inference(..., pixel_major=True)."""
    fn = ctx._lib.wsc_cam_unary_pm if pixel_major else ctx._lib.wsc_cam_unary
    check(fn(ctx.h, _ptr(cam_dev), B, C, h, w, H0, W0, float(bg_value), _ptr(unary_dev)))


def cam_sum_scales(ctx, cam_dev, n_images, n_scales, map_elems, out_dev):
    """out[b] = sum_s cam[b * n_scales + s] (make_cam.py:62-69 for equal-size scales; see include/wsscam.h)."""
    check(ctx._lib.wsc_cam_sum_scales(ctx.h, _ptr(cam_dev), int(n_images), int(n_scales), int(map_elems), _ptr(out_dev)))


def bilinear_resize(ctx, src_dev, C, h, w, dst_dev, H, W):
    check(ctx._lib.wsc_bilinear_resize(ctx.h, _ptr(src_dev), C, h, w, _ptr(dst_dev), H, W))


class Crf:
    """wsc_crf: lattices of a batch of images (DenseCRF2D + addPairwise*)."""

    def __init__(self, ctx, rgb_dev, B, H, W, g_sxy, bi_sxy, bi_srgb):
        self.ctx = ctx
        self.B, self.H, self.W = B, H, W
        h = _vp()
        check(ctx._lib.wsc_crf_create(ctx.h, _ptr(rgb_dev), B, H, W, float(g_sxy), float(bi_sxy), float(bi_srgb),
                                      ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx._lib.wsc_crf_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lattice_sizes(self):
        vg = np.zeros(self.B, dtype=np.int32)
        vb = np.zeros(self.B, dtype=np.int32)
        check(self.ctx._lib.wsc_crf_lattice_sizes(self.ctx.h, self.h, vg.ctypes.data, vb.ctypes.data))
        return vg, vb

    def gaussian_on_chip(self, M):
        """True when the update kernel blurs the Gaussian lattice itself for an M-class inference (wsc_crf_gaussian_on_chip)."""
        return bool(self.ctx._lib.wsc_crf_gaussian_on_chip(self.h, int(M)))

    def inference(self, unary_dev, M, g_compat, bi_compat, n_iters, q_dev=None, argmax_dev=None, ctx=None, pixel_major=False):
        """ctx: the context (stream, workspace) to run the mean-field loop on; defaults to the one the
        lattices were built on.  When it differs, the caller orders the two with ctx.wait_for(build_ctx).
        pixel_major: unary_dev is [B][H*W][Mp] (cam_unary(..., pixel_major=True)), read in place."""
        run = ctx or self.ctx
        fn = self.ctx._lib.wsc_crf_inference_pm if pixel_major else self.ctx._lib.wsc_crf_inference
        check(fn(run.h, self.h, _ptr(unary_dev), M, float(g_compat), float(bi_compat), int(n_iters), _ptr(q_dev),
                 _ptr(argmax_dev)))


def _ptr_array(ptrs):
    """list of DeviceBuffer / int addresses / None -> ctypes array of void* (kept alive by the caller)."""
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = None if p is None else (p.ptr if hasattr(p, "ptr") else int(p))
    return arr


class CrfV:
    """wsc_crf_v: lattices of a RAGGED batch -- every image its own (H, W) and, at inference, its own class count.
    rgb_ptrs: one device pointer (DeviceBuffer or address) per image, uint8 [H_b][W_b][3]; sizes: [(H_b, W_b)]."""

    def __init__(self, ctx, rgb_ptrs, sizes, g_sxy, bi_sxy, bi_srgb):
        self.ctx = ctx
        self.B = len(rgb_ptrs)
        self.sizes = [(int(h), int(w)) for h, w in sizes]
        assert len(self.sizes) == self.B
        hw = np.ascontiguousarray(self.sizes, dtype=np.int32).reshape(self.B, 2)
        arr = _ptr_array(rgb_ptrs)
        h = _vp()
        check(ctx._lib.wsc_crf_v_create(ctx.h, arr, hw.ctypes.data, self.B, float(g_sxy), float(bi_sxy), float(bi_srgb),
                                        ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx._lib.wsc_crf_v_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_groups(self):
        return int(self.ctx._lib.wsc_crf_v_num_groups(self.h))

    def inference(self, unary_ptrs, Ms, g_compat, bi_compat, n_iters, q_ptrs=None, argmax_ptrs=None, ctx=None):
        """unary_ptrs[b]: float32 [M_b][H_b*W_b] on the device; q_ptrs / argmax_ptrs: per-image outputs (lists, or None)."""
        run = ctx or self.ctx
        m = np.ascontiguousarray(Ms, dtype=np.int32)
        assert len(unary_ptrs) == self.B and len(m) == self.B
        ua = _ptr_array(unary_ptrs)
        qa = _ptr_array(q_ptrs) if q_ptrs is not None else None
        aa = _ptr_array(argmax_ptrs) if argmax_ptrs is not None else None
        check(self.ctx._lib.wsc_crf_v_inference(run.h, self.h, ua, m.ctypes.data, float(g_compat), float(bi_compat), int(n_iters),
                                                qa, aa))


def rw_propagate(ctx, x_dev, edge_dev, K, h, w, dirs, path_start, path_yx, beta, n_steps, rw_dev=None):
    """wsc_rw_propagate: x_dev [K][h][w], edge_dev [h][w] device buffers -> rw_dev [K][h][w] (allocated if None)."""
    import numpy as np

    dirs = np.ascontiguousarray(dirs, dtype=np.int32)
    path_start = np.ascontiguousarray(path_start, dtype=np.int32)
    path_yx = np.ascontiguousarray(path_yx, dtype=np.int32)
    if rw_dev is None:
        rw_dev = ctx.alloc(K * h * w * 4)
    check(ctx._lib.wsc_rw_propagate(ctx.h, _ptr(x_dev), _ptr(edge_dev), K, h, w, dirs.ctypes.data_as(_vp),
                                    path_start.ctypes.data_as(_vp), path_yx.ctypes.data_as(_vp), int(dirs.shape[0]),
                                    float(beta), int(n_steps), _ptr(rw_dev)))
    return rw_dev


def rw_propagate_batch(ctx, x_dev, edge_dev, Ks, hs, ws, dirs, path_start, path_yx, beta, n_steps, rw_dev=None):
    """wsc_rw_propagate_batch: packed [K_b][h_b][w_b] blocks / [h_b][w_b] edge maps of several images."""
    import numpy as np

    Ks, hs, ws = (np.ascontiguousarray(v, dtype=np.int32) for v in (Ks, hs, ws))
    dirs = np.ascontiguousarray(dirs, dtype=np.int32)
    path_start = np.ascontiguousarray(path_start, dtype=np.int32)
    path_yx = np.ascontiguousarray(path_yx, dtype=np.int32)
    if rw_dev is None:
        rw_dev = ctx.alloc(int((Ks.astype(np.int64) * hs * ws).sum()) * 4)
    check(ctx._lib.wsc_rw_propagate_batch(ctx.h, int(len(Ks)), Ks.ctypes.data_as(_vp), hs.ctypes.data_as(_vp),
                                          ws.ctypes.data_as(_vp), _ptr(x_dev), _ptr(edge_dev), dirs.ctypes.data_as(_vp),
                                          path_start.ctypes.data_as(_vp), path_yx.ctypes.data_as(_vp),
                                          int(dirs.shape[0]), float(beta), int(n_steps), _ptr(rw_dev)))
    return rw_dev



# ---- HistoSegNet post-processing (csrc/hsn.hip) -------------------------------------------------------------------
def hsn_gradcam_post(ctx, cams_nhwc_dev, B, h, w, C, S, gate_dev, out_dev, out_channels=0, out_first=0):
    check(ctx._lib.wsc_hsn_gradcam_post(ctx.h, _ptr(cams_nhwc_dev), B, h, w, C, S, _ptr(gate_dev), _ptr(out_dev),
                                        int(out_channels), int(out_first)))


def hsn_voc_background(ctx, Hbg_dev, B, Cb, N, y_dev, Ctot):
    check(ctx._lib.wsc_hsn_voc_background(ctx.h, _ptr(Hbg_dev), B, Cb, N, _ptr(y_dev), Ctot))


def hsn_class_mass(ctx, maps_dev, n_maps, N, mass_dev):
    check(ctx._lib.wsc_hsn_class_mass(ctx.h, _ptr(maps_dev), n_maps, N, _ptr(mass_dev)))


def hsn_background(ctx, rgb_dev, B, H, W, bg_dev, out_hw=None):
    Ho, Wo = (H, W) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    check(ctx._lib.wsc_hsn_background(ctx.h, _ptr(rgb_dev), B, H, W, Ho, Wo, _ptr(bg_dev)))


def cam_adp_modify(ctx, cam_dev, B, n_sc, C, hw, bg_dev, mode, use, adipose, exc, out_dev):
    """wsc_cam_adp_modify: common_cam.py:31-92 on the device; -> number of output channels (len(use) + 1 + mode)."""
    us = np.ascontiguousarray(use, dtype=np.int32)
    ad = np.ascontiguousarray(adipose, dtype=np.int32)
    ex = np.ascontiguousarray(exc if exc is not None else [], dtype=np.int32)
    check(ctx._lib.wsc_cam_adp_modify(ctx.h, _ptr(cam_dev), int(B), int(n_sc), int(C), int(hw), _ptr(bg_dev), int(mode),
                                      us.ctypes.data, len(us), ad.ctypes.data, len(ad), ex.ctypes.data if len(ex) else None,
                                      len(ex), _ptr(out_dev)))
    return len(us) + 1 + int(mode)


def hsn_cs_gradcam(ctx, H_dev, B, C_all, N, bg_dev, src_of_valid, bg_ind, other_ind, exception_inds, adipose_src, cs_dev,
                   y_dev, mass_dev):
    sv = np.ascontiguousarray(src_of_valid, dtype=np.int32)
    ex = np.ascontiguousarray(exception_inds, dtype=np.int32)
    ad = np.ascontiguousarray(adipose_src if adipose_src is not None else [], dtype=np.int32)
    check(ctx._lib.wsc_hsn_cs_gradcam(ctx.h, _ptr(H_dev), B, C_all, N, _ptr(bg_dev), sv.ctypes.data, len(sv), int(bg_ind),
                                      int(other_ind), ex.ctypes.data if len(ex) else None, len(ex),
                                      ad.ctypes.data if len(ad) else None, len(ad), _ptr(cs_dev), _ptr(y_dev), _ptr(mass_dev)))


def hsn_gather_unary(ctx, maps_dev, chan_off, N, unary_dev):
    co = np.ascontiguousarray(chan_off, dtype=np.int64)
    check(ctx._lib.wsc_hsn_gather_unary(ctx.h, _ptr(maps_dev), co.ctypes.data, len(co), N, _ptr(unary_dev)))


def msf_input_u8(ctx, images_dev, sizes, offsets, S, mean, std, x_dev, pre_div255=False, pair=True):
    """wsc_msf_input_u8: decoded uint8 images (packed HWC blocks at byte `offsets`) -> float32 network input."""
    B = len(sizes)
    size_hw = np.ascontiguousarray(sizes, dtype=np.int32).reshape(B, 2)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    m = np.ascontiguousarray(mean, dtype=np.float32)
    sd = np.ascontiguousarray(std, dtype=np.float32)
    check(ctx._lib.wsc_msf_input_u8(ctx.h, _ptr(images_dev), B, size_hw.ctypes.data, off.ctypes.data, int(S), m.ctypes.data,
                                    sd.ctypes.data, int(bool(pre_div255)), int(bool(pair)), _ptr(x_dev)))


def resize_u8(ctx, images_dev, sizes, offsets, out_hw, out_dev):
    """wsc_resize_u8: cv2.resize (INTER_LINEAR, 8-bit fixed point) of packed uint8 HWC images -> uint8 [B][OH][OW][3]."""
    B = len(sizes)
    size_hw = np.ascontiguousarray(sizes, dtype=np.int32).reshape(B, 2)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    check(ctx._lib.wsc_resize_u8(ctx.h, _ptr(images_dev), B, size_hw.ctypes.data, off.ctypes.data, int(out_hw[0]), int(out_hw[1]),
                                 _ptr(out_dev)))


def label_unary_from_cam(ctx, highres_dev, B, K, N, thres, gt_prob, unary_dev, labels_dev=None):
    check(ctx._lib.wsc_label_unary_from_cam(ctx.h, _ptr(highres_dev), B, K, N, float(thres), float(gt_prob), _ptr(unary_dev),
                                            _ptr(labels_dev)))


def ir_label_combine(ctx, fg_pred_dev, bg_pred_dev, keys, N, conf_dev):
    k = np.ascontiguousarray(keys, dtype=np.int32)
    B, M = k.shape
    check(ctx._lib.wsc_ir_label_combine(ctx.h, _ptr(fg_pred_dev), _ptr(bg_pred_dev), k.ctypes.data, B, M, N, _ptr(conf_dev)))
