#!/usr/bin/env python3
"""tests/golden/exr_face_hashes.json: FNV-1a-64 of every cubemap face (float32 RGB, row-major, as Bitmap::loadEXR leaves it, bitmap.cpp:238-264) of
the scenes that ship EXR environments, decoded by oracle/exr_piz_reader.py -- the INDEPENDENT reader written from the OpenEXR file-layout / PIZ
description, not by the product's fray_amd/csrc/host_exr.cpp, whose output tests/test_host_scene.py then holds against these hashes.  (The
reference decodes through the OpenEXR library, which this image lacks: two implementations that share only the specification are the strongest pin
available here; both also reproduce, bit for bit, synthetic PIZ files of odd sizes and small value ranges written by oracle/exr_piz_writer.py.)"""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import exr_piz_reader  # noqa: E402


def environment_folder(scene_path):
    """The `folder` of the scene's CubemapEnvironment block, resolved like the reference resolves file names: relative to the scene file."""
    text = open(scene_path).read()
    m = re.search(r'CubemapEnvironment(?:\s+\w+)?\s*\{[^}]*?folder\s+"([^"]+)"', text, re.S)
    return os.path.join(os.path.dirname(scene_path), m.group(1)) if m else None


if __name__ == "__main__":
    out = {"_comment": "decoded by oracle/exr_piz_reader.py (independent of fray_amd/csrc/host_exr.cpp); made by oracle/make_exr_hashes.py"}
    for scene in ("forest.fray", "hw10/bokeh.fray", "hw9/axe_test.fray"):
        folder = environment_folder(os.path.join(ROOT, "scenes", scene))
        if not folder or not os.path.isdir(folder):
            continue
        faces = []
        for n in ("negx", "negy", "negz", "posx", "posy", "posz"):
            img = exr_piz_reader.read_rgb(os.path.join(folder, n + ".exr"))
            faces.append({"face": n, "w": img.shape[1], "h": img.shape[0], "fnv": exr_piz_reader.fnv1a64(img), "mean": float(img.mean())})
        out[scene] = faces
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "exr_face_hashes.json"), "w"), indent=1)
    print({k: [f["fnv"] for f in v] for k, v in out.items() if k != "_comment"})
