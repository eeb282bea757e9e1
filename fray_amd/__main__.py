"""File-only front end over the C ABI (the reference's main() opens an SDL window instead,
src/main.cpp:494-530):  python -m fray_amd scene.fray -o out.bmp [--width W --height H --spp N]"""
import argparse
import sys
import time

from . import Scene, lib


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m fray_amd")
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default="fray_0000.bmp")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int, help="pathsPerPixel (gi scenes) / numSamples (dof scenes)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    s = Scene.parseScene(a.scene)
    if a.width:
        s.settings.frameWidth = a.width
    if a.height:
        s.settings.frameHeight = a.height
    if a.spp:
        if s.settings.gi:
            s.settings.numPaths = a.spp
        elif s.camera.dof:
            s.camera.numDOFSamples = a.spp
    s.beginRender(a.device)
    t0 = time.time()
    img, st = s.render(seed=a.seed)
    print("Render took %.2fs (%d x %d, %d spp, kernels %.1f ms)" % (time.time() - t0, img.shape[1], img.shape[0], s.samples_per_pixel(), st["ms_kernels"]))
    rc = lib.frayhip_save_bmp(a.output.encode(), img.ctypes.data, img.shape[1], img.shape[0])
    if rc:
        print(lib.frayhip_last_error().decode(), file=sys.stderr)
        return 1
    print("wrote", a.output)
    return 0


if __name__ == "__main__":
    sys.exit(main())
