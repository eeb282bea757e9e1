"""Identity of the device code a measurement belongs to: SHA-1 over the kernel / library sources and the build
recipe.  bench.py prints counter-derived figures from profiles/ only when the profile carries this hash."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha1()
    # exact suffixes only: an editor's or a compiler's leftover next to the sources (x.hpp~, y.hip.tmp) must not change the identity
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "fray_amd", "csrc", "*")) if f.endswith((".hip", ".hpp", ".h", ".cpp")))
    files += [os.path.join(ROOT, "include", "frayhip.h"), os.path.join(ROOT, "Makefile")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
