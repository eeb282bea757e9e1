"""ctypes loader for the CPU oracle (oracle/libfray_oracle.so).  TEST INFRASTRUCTURE ONLY: imported
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by fray_amd/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libfray_oracle.so")


def load(abi):
    lib = C.CDLL(_PATH)
    P = C.POINTER
    lib.fray_oracle_render.restype = C.c_int
    lib.fray_oracle_render.argtypes = [P(abi.SceneDesc), P(abi.Frame), C.c_void_p, C.c_void_p, C.c_void_p, P(abi.Stats), C.c_int]
    lib.fray_oracle_fnv1a64.restype = C.c_uint64
    lib.fray_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
    lib.fray_oracle_sample_seed.restype = C.c_uint32
    lib.fray_oracle_sample_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.fray_oracle_rng_words.restype = None
    lib.fray_oracle_rng_words.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
    lib.fray_oracle_rng_stream.restype = None
    lib.fray_oracle_rng_stream.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.fray_oracle_probe.restype = C.c_int
    lib.fray_oracle_probe.argtypes = [P(abi.SceneDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fray_oracle_camera_ray.restype = None
    lib.fray_oracle_camera_ray.argtypes = [P(abi.SceneDesc), C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    return lib


class Oracle:
    def __init__(self, abi):
        self.abi = abi
        self.lib = load(abi)

    def fnv(self, arr):
        arr = np.ascontiguousarray(arr)
        return "%016x" % self.lib.fray_oracle_fnv1a64(arr.ctypes.data, arr.nbytes)

    def render(self, desc, mode, seed=42, bucket_first=0, bucket_stride=1, threads=8):
        abi = self.abi
        W, H = desc.settings.frameWidth, desc.settings.frameHeight
        fr = abi.Frame(mode=mode, seed=seed, bucket_first=bucket_first, bucket_stride=bucket_stride, spp_chunk=0, flags=0)
        st = abi.Stats()
        if mode == abi.MODE_PRIMARY_ID:
            ids = np.full((H, W), -9, np.int32)
            dist = np.zeros((H, W), np.float64)
            rc = self.lib.fray_oracle_render(C.byref(desc), C.byref(fr), None, ids.ctypes.data, dist.ctypes.data, C.byref(st), threads)
            assert rc == 0, rc
            return ids, dist, st.as_dict()
        rgb = np.zeros((H, W, 3), np.float32)
        rc = self.lib.fray_oracle_render(C.byref(desc), C.byref(fr), rgb.ctypes.data, None, None, C.byref(st), threads)
        assert rc == 0, rc
        return rgb, st.as_dict()
