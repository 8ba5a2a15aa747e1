"""raytracinggpu_amd -- MI355X-native render path for souhhcong/RaytracingGPU.

The product is libraytrace_hip.so (hand-written HIP for gfx950 behind the C-ABI of
include/raytrace_hip.h) plus the C++ host API of include/raytracer.hpp.  This
Python package is plumbing for tests and bench.py: a ctypes binding and the scene
presets of the reference programs.
"""
from . import _capi, scenes  # noqa: F401  (tiling imports torch: import it explicitly where needed)
from ._capi import Context, MultiContext, PinnedArray, RtError, camera_basis, device_count, interleaved_rows, make_params, make_pose  # noqa: F401
