"""khronos_amd — MI355X-native active-window volumetric fusion path of MIT-SPARK/Khronos.

The product is the HIP library (csrc/ -> lib/libkhronos_amd.so) behind the C ABI in
include/khronos_amd.h; this package holds the ctypes plumbing used by tests and bench.py and the
host-side mirror of the reference's khronos::ActiveWindow interface.
"""
from .capi import FusionContext, KhronosAmdError, RayVerificator, default_config, load_library  # noqa: F401
