"""openkitchen_amd -- MI355X-native batched implementation of OpenKitchen's Environment step path.

The product is libokenv.so (hand-written HIP kernels for gfx950 behind the C ABI of include/okenv.h, plus
the C++ Environment/Agent/CollisionChecker facade of include/Environment/).  This package is the Python
host side: it builds and loads that library and exposes typed wrappers.  Nothing here computes on the CPU
in its place.
"""
from . import _capi as capi  # noqa: F401
from .buildlib import build  # noqa: F401
from .env import BatchedEnvironment, Track, debug_sincos, default_ray_fan, track_path  # noqa: F401

__all__ = ["BatchedEnvironment", "Track", "build", "capi", "debug_sincos", "default_ray_fan", "track_path"]
