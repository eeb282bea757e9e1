#!/usr/bin/env python3
"""Primary-ray known answers measured on oracle/_ref (the reference's own parser, camera, transform,
geometry, KD and light object code; oracle/ref_glue.cpp ref_primary) for tests/golden/primary_hashes.json:

  * "cases" was measured by the survey on the complete reference (built there with stand-in SDL / OpenEXR
    headers).  This script measures the same eight cases again on oracle/_ref -- eleven reference
    translation units compiled exactly as they are, no stand-in header anywhere -- and FAILS if any hash
    differs; every case that agrees is stamped "reproduced_on_ref": true, so the eight known answers rest
    on the clean build too.  The survey's numbers themselves are never rewritten.
  * "cases_ref": five more cases (one per remaining scene file) measured only here.

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
    """FNV-1a-64 of the array's bytes (offset basis 14695981039346656037, prime 1099511628211): the checksum SURVEY 8c defines.  The byte loop runs in
    C when the oracle's helper library is built (134 MB for the 4096x4096 case); the pure-Python loop below is the definition."""
    so = os.path.join(HERE, "libfray_oracle.so")
    a = np.ascontiguousarray(a)
    if os.path.exists(so):
        lib = C.CDLL(so)
        lib.fray_oracle_fnv1a64.restype = C.c_uint64
        lib.fray_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        return "%016x" % lib.fray_oracle_fnv1a64(a.ctypes.data, a.nbytes)
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
        def measure(c):      # one process per scene: the reference's `scene` is a process-wide singleton
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(x) for x in c], check=True, capture_output=True, text=True)
            line = [l for l in r.stderr.splitlines() if l.startswith("RESULT ")][-1]
            return json.loads(line[7:])

        g = json.load(open(PATH))
        # the survey's eight cases again, on the clean build
        differing = []
        for case in g["cases"]:
            got = measure((case["scene"], case["w"], case["h"]))
            same = all(got[k] == case[k] for k in ("hits", "id", "dist"))
            print(("reproduced " if same else "DIFFERS    ") + json.dumps(got))
            if same:
                case["reproduced_on_ref"] = True
            else:
                case.pop("reproduced_on_ref", None)
                differing.append((case, got))
        out = []
        for c in CASES:
            out.append(measure(c))
            print(out[-1])
        g["_comment_ref"] = ("cases_ref: the same kind of hashes measured on oracle/_ref (reference object code for parser / camera / transforms / geometry / KD / "
                             "lights, oracle/ref_glue.cpp ref_primary) by oracle/make_primary_hashes.py; \"reproduced_on_ref\" in \"cases\": that script measured "
                             "the survey's case again on oracle/_ref and got the survey's hits and both hashes")
        g["cases_ref"] = out
        g["reproduced_on_ref"] = not differing
        json.dump(g, open(PATH, "w"), indent=1)
        if differing:
            sys.exit("oracle/_ref does not reproduce the survey's hashes: %s" % differing)
