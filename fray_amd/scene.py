"""Host-side mirror of the reference's Scene / render() interface over the C ABI."""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRAYHIP_LIB: A/B-test another build of the same library (development only)
_LIB_PATH = os.environ.get("FRAYHIP_LIB") or os.path.join(_HERE, "libfrayhip.so")


class FrayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("frayhip error %d: %s" % (code, msg))
        self.code = code


def _load():
    # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no versioned SONAME).
    # A process must not end up with two HIP runtimes: whichever is loaded second finds no GPU and
    # stream handles do not carry over.  Loading torch first puts its runtime in the global symbol
    # scope, and libfrayhip.so then binds to that same runtime.  (Hosts that never use torch --
    # e.g. the reference's C++ main() -- just get /opt/rocm's runtime.)
    if "torch" not in sys.modules and not os.environ.get("FRAYHIP_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            "fray_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make`). There is no CPU fallback." % _LIB_PATH)
    return abi.bind(C.CDLL(_LIB_PATH))


lib = _load()


def _check(rc):
    if rc != 0:
        raise FrayError(rc, (lib.frayhip_last_error() or b"").decode("utf-8", "replace"))


def render_info():
    return {"library": _LIB_PATH, "abi_version": lib.frayhip_abi_version()}


class Scene:
    """`Scene scene` of the reference (scene.h:280-299).

    parseScene()  -> Scene::parseScene + the beginRender work that needs no GPU (KD build)
    settings / camera -> mutable GlobalSettings / Camera records, edited before beginRender() the way
                     the reference's callers edit scene.settings.* after parsing
    beginRender() -> uploads the flattened scene to the current GPU
    render()      -> render() (main.cpp:373): returns vfb as float32 [H, W, 3]
    """

    def __init__(self):
        self._hs = C.c_void_p()
        self._dev = C.c_void_p()
        self.desc = None

    @classmethod
    def parseScene(cls, path):
        s = cls()
        _check(lib.frayhip_scene_parse(os.fspath(path).encode(), C.byref(s._hs)))
        s.desc = lib.frayhip_host_scene_desc(s._hs).contents
        return s

    @property
    def settings(self):
        return self.desc.settings

    @property
    def camera(self):
        return self.desc.camera

    @property
    def frame_size(self):
        return self.desc.settings.frameWidth, self.desc.settings.frameHeight

    def samples_per_pixel(self):
        """main.cpp:395-400"""
        spp = 5 if self.settings.wantAA else 1
        if self.camera.dof:
            spp = max(spp, self.camera.numDOFSamples)
        if self.settings.gi:
            spp = max(spp, self.settings.numPaths)
        return spp

    def beginRender(self, device=None):
        if device is not None:
            _check(lib.frayhip_init(int(device)))
        self.endRender()
        _check(lib.frayhip_scene_create(C.byref(self.desc), C.byref(self._dev)))
        return self

    def beginFrame(self):
        """Pushes the current settings / camera records to the uploaded scene (Scene::beginFrame: the
        reference re-derives the camera every frame, so callers may move it between render() calls)."""
        self._need_dev()
        _check(lib.frayhip_scene_set_view(self._dev, C.byref(self.desc.camera), C.byref(self.desc.settings)))
        return self

    def endRender(self):
        if self._dev:
            lib.frayhip_scene_destroy(self._dev)
            self._dev = C.c_void_p()

    def set_option(self, name, value):
        """frayhip_scene_set_option: "pt_lanes" (1..4 batches in flight), "pt_budget_mib" (queue memory), "speculate_fans" (0 / 1), "fp_contract" (0 / 1: relaxed arithmetic
        for path-traced rays after a sample's first closest hit, include/frayhip.h)."""
        self._need_dev()
        _check(lib.frayhip_scene_set_option(self._dev, name.encode(), int(value)))
        return self

    def get_option(self, name):
        """frayhip_scene_get_option: an option's value, or a figure of the last frame ("fans_filed", "fan_children", "fan_children_looked_up", "fans_given_up")."""
        self._need_dev()
        v = C.c_int64(0)
        _check(lib.frayhip_scene_get_option(self._dev, name.encode(), C.byref(v)))
        return int(v.value)

    def _frame(self, mode, seed, bucket_first, bucket_stride, spp_chunk, stats):
        return abi.Frame(mode=mode, seed=seed, bucket_first=bucket_first, bucket_stride=bucket_stride,
                         spp_chunk=spp_chunk, flags=abi.FRAME_STATS if stats else 0)

    def _need_dev(self):
        if not self._dev:
            raise FrayError(abi.E_ARG, "Scene.beginRender() has not been called")

    def render(self, seed=42, bucket_first=0, bucket_stride=1, spp_chunk=0, stats=False, out=None):
        """Full render; returns (vfb, stats dict)."""
        self._need_dev()
        W, H = self.frame_size
        rgb = out if out is not None else np.zeros((H, W, 3), np.float32)
        st = abi.Stats()
        fr = self._frame(abi.MODE_RENDER, seed, bucket_first, bucket_stride, spp_chunk, stats)
        _check(lib.frayhip_render(self._dev, C.byref(fr), rgb.ctypes.data, None, None, C.byref(st)))
        return rgb, st.as_dict()

    def primary_hits(self, bucket_first=0, bucket_stride=1, stats=False):
        """Closest hit of the camera ray through every integer pixel: (ids int32 [H,W], dist f64 [H,W], stats)."""
        self._need_dev()
        W, H = self.frame_size
        ids = np.full((H, W), -9, np.int32)
        dist = np.zeros((H, W), np.float64)
        st = abi.Stats()
        fr = self._frame(abi.MODE_PRIMARY_ID, 0, bucket_first, bucket_stride, 0, stats)
        _check(lib.frayhip_render(self._dev, C.byref(fr), None, ids.ctypes.data, dist.ctypes.data, C.byref(st)))
        return ids, dist, st.as_dict()

    def render_device(self, d_rgb_ptr, seed=42, bucket_first=0, bucket_stride=1, spp_chunk=0, stats=False,
                      stream=None, mode=abi.MODE_RENDER, d_id_ptr=None, d_dist_ptr=None):
        """Render into caller-owned device memory (e.g. torch tensors' data_ptr())."""
        self._need_dev()
        st = abi.Stats()
        fr = self._frame(mode, seed, bucket_first, bucket_stride, spp_chunk, stats)
        _check(lib.frayhip_render_device(self._dev, C.byref(fr), d_rgb_ptr, d_id_ptr, d_dist_ptr, stream, C.byref(st)))
        return st.as_dict()

    def close(self):
        self.endRender()
        if self._hs:
            lib.frayhip_host_scene_free(self._hs)
            self._hs = C.c_void_p()
            self.desc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
