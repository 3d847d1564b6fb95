"""ultrasound_modeling_amd: MI355X-native (gfx950) training path for the ResNeSt/UNet ultrasound segmenter of
silverlight6/Ultrasound_Modeling.  Python host code (this package) mirrors the reference's module surface
(ResNest.py, Decoder.py, VisionTransformer.py, TBI_ResNest.py, MainParallel.py) and calls hand-written HIP kernels
through the C ABI of ``libusseg_hip.so`` (include/usseg.h).  There is no CPU fallback."""
from ._lib import LIB_PATH, UssegError, load  # noqa: F401

__all__ = ["LIB_PATH", "UssegError", "load"]
