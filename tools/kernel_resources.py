"""Table of the compiler's per-kernel resource report (hipcc -Rpass-analysis=kernel-resource-usage, written by
the Makefile next to each variant object): registers, spills, scratch, LDS, occupancy.

usage: python tools/kernel_resources.py fray_amd/csrc/variant*.resources.txt [> profiles/rNN_kernel_resources.txt]
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return [re.sub(r"\(.*", "", l).replace("void ", "") for l in out.splitlines()]
    except (OSError, subprocess.SubprocessError):
        return names


def main():
    rows = []
    for path in sys.argv[1:]:
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z /\[\]]+?): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["name"] = n
    seen, uniq = set(), []
    for r in rows:                      # the plain (non-template) kernels appear once per variant object
        key = tuple(sorted(r.items()))
        if key not in seen:
            seen.add(key)
            uniq.append(r)
    rows = uniq
    cols = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("VGPRs Spill", "vgpr_spill"), ("SGPRs Spill", "sgpr_spill"),
            ("ScratchSize [bytes/lane]", "scratch_B"), ("LDS Size [bytes/block]", "lds_B"), ("Occupancy [waves/SIMD]", "waves/SIMD")]
    print("%-34s" % "kernel" + "".join("%12s" % c[1] for c in cols))
    for r in sorted(rows, key=lambda r: r["name"]):
        print("%-34s" % r["name"] + "".join("%12s" % r.get(c[0], "-") for c in cols))


if __name__ == "__main__":
    main()
