"""raytracer_amd: MI355X (gfx950) implementation of the zkkv/raytracer render hot path.

The product is the shared library raytracer_amd/lib/librt_amd.so (HIP kernels behind the C ABI of
include/rt_amd.h) and the C++ host mirror of the reference's interfaces under raytracer_amd/host/.
The Python modules here (api, scenes) are harness plumbing for tests and bench.py."""
from . import api, scenes  # noqa: F401

__all__ = ["api", "scenes"]
