#!/usr/bin/env python3
"""Generates tests/golden/ref_parse.json: what the reference's OWN parser object code (oracle/_ref: scene.o with every fillProperties, the OBJ / BMP
loaders, Transform) made of (a) the parser edge-case texts of tests/parser_cases.py -- comments, quotes, singleton blocks, transform order, Layered
lines with forward references, randfloat / randint macros, OBJ dummy indices, ... -- and (b) every scene file in the repository.  The dump format is
oracle/ref_dump.cpp's; tests/test_host_scene.py compares the PRODUCT's parse of the same files (oracle/scene_dump.py) with these, token by token.

    python oracle/make_parse_golden.py        # needs /root/reference mounted (make ref)
One subprocess per scene: the reference's `scene` is a process-wide singleton.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "ref_parse.json")
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHIPPED = ["boxed.fray", "zaphod.fray", "cornell_box.fray", "forest.fray", "smallpt.fray", "hw9/dragon.fray", "hw12/sphtri.fray", "hw10/bokeh.fray",
           "hw9/axe_test.fray", "hw9/nonconvex.fray", "../tests/scenes/csg_nested.fray", "../tests/scenes/csg_deep.fray", "../tests/scenes/whitebox.fray",
           "../tests/scenes/fuzz1009/scene.fray", "../tests/scenes/fuzz3001/scene.fray", "../tests/scenes/fuzz3004/scene.fray", "../tests/scenes/fuzz3010/scene.fray"]


def worker(path):
    lib = C.CDLL(os.path.join(HERE, "_ref", "libfray_ref.so"))
    lib.ref_parse.argtypes = [C.c_char_p]
    rc = lib.ref_parse(path.encode())
    if rc:
        sys.stderr.write("REFDUMP-FAILED %d\n" % rc)
        return
    buf = C.create_string_buffer(1 << 22)
    n = lib.ref_dump_scene(buf, len(buf))
    assert n > 0
    sys.stderr.write("REFDUMP-BEGIN\n" + buf.value.decode() + "REFDUMP-END\n")


def ref_dump(path):
    """The reference's dump of one scene file as a list of lines, or None when its parser rejects the file."""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), path], capture_output=True, text=True, timeout=600)
    if "REFDUMP-BEGIN\n" not in r.stderr:
        return None
    return r.stderr.split("REFDUMP-BEGIN\n", 1)[1].split("REFDUMP-END\n", 1)[0].splitlines()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1])
    else:
        from parser_cases import CASES, write_case
        out = {"_comment": "oracle/make_parse_golden.py: the reference's parser object code (oracle/_ref) on tests/parser_cases.py's texts and on the repository's "
                           "scene files; format of oracle/ref_dump.cpp; null = the reference rejects the text", "cases": {}, "scenes": {}}
        for name in sorted(CASES):
            with tempfile.TemporaryDirectory() as tmp:
                out["cases"][name] = ref_dump(write_case(name, tmp))
            print("case", name, "rejected" if out["cases"][name] is None else "%d lines" % len(out["cases"][name]))
        for name in SHIPPED:
            out["scenes"][name] = ref_dump(os.path.join(ROOT, "scenes", name))
            print("scene", name, "rejected" if out["scenes"][name] is None else "%d lines" % len(out["scenes"][name]))
        json.dump(out, open(OUT, "w"), indent=0)
