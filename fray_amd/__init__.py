"""fray_amd -- MI355X-native renderer for fray's per-pixel ray-trace hot path.

Python is only the plumbing above the C ABI (include/frayhip.h, fray_amd/libfrayhip.so): it mirrors
the reference's `Scene` interface (src/scene.h:280-299: parseScene / settings / camera /
beginRender) and `render()` (src/main.cpp:373).  All rendering happens in the HIP library; there is
no CPU fallback -- if the library is missing or no GPU is present the calls fail loudly.
"""
from .scene import Scene, FrayError, lib, render_info  # noqa: F401
from . import abi  # noqa: F401

__all__ = ["Scene", "FrayError", "lib", "abi", "render_info"]
