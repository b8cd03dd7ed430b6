"""openclraytracer_amd -- MI355X (gfx950) ray-intersection engine behind the OpenCLRayTracer API.

Product code = the HIP/C++ sources under ``csrc/`` built into ``librtmi.so`` (C ABI: include/rtmi.h).
``capi`` is the ctypes binding used by tests/ and bench.py.  No CPU fallback exists.
"""
from . import capi  # noqa: F401

__all__ = ['capi']
