"""wsscam -- host-side mirror of the wsss-analysis CAM pseudo-label interfaces on top of
libwsscam (HIP, gfx950).  Sub-packages keep the reference's module names:

    wsscam.step.make_cam.run(args)            03b_irn/step/make_cam.py
    wsscam.net.resnet50_cam.CAM               03b_irn/net/resnet50_cam.py
    wsscam.net.vgg16_cam.CAM / m7_cam.CAM     03b_irn/net/vgg16_cam.py, m7_cam.py
    wsscam.misc.imutils / torchutils          03b_irn/misc (not in the reference tree)
    wsscam.hsn.utilities.dcrf_process         03c_hsn/utilities.py:399-445
    wsscam.cues.utilities.grad_cam ...        02_cues/utilities.py

There is no CPU fallback: every compute entry point goes through the C ABI of
include/wsscam.h and raises WscError when the library or a gfx950 device is missing.
"""
from . import _lib  # noqa: F401
from ._lib import WscError  # noqa: F401

__version__ = "0.1.0"
