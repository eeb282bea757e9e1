#!/usr/bin/env python3
"""tests/golden/exr_face_hashes.json: FNV-1a-64 of every decoded cubemap face (float32 RGB, row-major) of the scenes
that ship EXR environments, as THIS project's decoder (fray_amd/csrc/host_exr.cpp) produces them.  Regression pins:
no OpenEXR exists in this image to decode the files independently (DESIGN.md section 2 says what that leaves open)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FRAYHIP_NO_TORCH", "1")
import fray_amd  # noqa: E402
from fray_amd import abi  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

if __name__ == "__main__":
    orc = Oracle(abi)
    out = {}
    for scene in ("forest.fray", "hw10/bokeh.fray", "hw9/axe_test.fray"):
        s = fray_amd.Scene.parseScene(os.path.join(ROOT, "scenes", scene))
        e = s.desc.environment
        if not (e.present and e.loaded):
            continue
        tex = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texels,))
        out[scene] = [{"face": n, "w": e.width[f], "h": e.height[f],
                       "fnv": orc.fnv(np.ascontiguousarray(tex[e.texel_offset[f]:e.texel_offset[f] + e.width[f] * e.height[f] * 3])),
                       "mean": float(tex[e.texel_offset[f]:e.texel_offset[f] + e.width[f] * e.height[f] * 3].mean())}
                      for f, n in enumerate(("negx", "negy", "negz", "posx", "posy", "posz"))]
        s.close()
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "exr_face_hashes.json"), "w"), indent=1)
    print({k: [f["fnv"] for f in v] for k, v in out.items()})
