"""Identity of the device code a measurement belongs to: SHA-1 over the kernel / library sources and the build
recipe.  bench.py prints counter-derived figures from profiles/ only when the profile carries this hash."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "fray_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "fray_amd", "csrc", "*.cpp")) +
                   [os.path.join(ROOT, "include", "frayhip.h"), os.path.join(ROOT, "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
