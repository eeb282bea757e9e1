#!/usr/bin/env python3
"""Adds primary-ray known answers measured on oracle/_ref (the reference's own camera, transform,
geometry, KD and light object code; oracle/ref_glue.cpp ref_primary) to
tests/golden/primary_hashes.json under "cases_ref".  The "cases" list in that file was measured
by the survey on the complete unmodified reference; this script does not touch it.

    python oracle/make_primary_hashes.py        # needs /root/reference mounted (make ref)
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PATH = os.path.join(ROOT, "tests", "golden", "primary_hashes.json")
CASES = [("hw10/bokeh.fray", 640, 480), ("hw9/axe_test.fray", 640, 480), ("hw9/nonconvex.fray", 640, 480),
         ("hw12/sphtri.fray", 640, 480), ("boxed.fray", 97, 61)]


def fnv(a):
    h = 14695981039346656037
    for b in a.tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def worker(scene, W, H):
    lib = C.CDLL(os.path.join(HERE, "_ref", "libfray_ref.so"))
    lib.ref_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.ref_primary.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.ref_load(os.path.join(ROOT, "scenes", scene).encode(), W, H, b"wantAA=0") == 0
    ids, dist = np.zeros(W * H, np.int32), np.zeros(W * H, np.float64)
    lib.ref_primary(ids.ctypes.data, dist.ctypes.data)
    sys.stderr.write("RESULT " + json.dumps({"scene": scene, "w": W, "h": H, "hits": int((ids != -1).sum()), "id": fnv(ids), "dist": fnv(dist)}) + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    else:
        out = []
        for c in CASES:      # one process per scene: the reference's `scene` is a process-wide singleton
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(x) for x in c], check=True, capture_output=True, text=True)
            line = [l for l in r.stderr.splitlines() if l.startswith("RESULT ")][-1]
            out.append(json.loads(line[7:]))
            print(out[-1])
        g = json.load(open(PATH))
        g["_comment_ref"] = ("cases_ref: the same hashes measured on oracle/_ref (reference object code for camera / transforms / geometry / KD / "
                             "lights, oracle/ref_glue.cpp ref_primary) by oracle/make_primary_hashes.py")
        g["cases_ref"] = out
        json.dump(g, open(PATH, "w"), indent=1)
