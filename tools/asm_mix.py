"""Static instruction mix of one kernel in a device assembly listing (hipcc ... --offload-device-only -S): how many
instructions of each class the kernel's TEXT holds (not how many execute).  Used to compare builds of the same source
(e.g. -ffp-contract=off against =fast) without a GPU.

usage: python tools/asm_mix.py listing.s kernel-name-substring"""
import collections
import re
import sys


def kernel_text(path, needle):
    out, on = [], False
    for line in open(path, errors="replace"):
        if re.match(r"^[_A-Za-z][\w.$]*:", line):
            name = line.split(":")[0]
            if on and not name.startswith(".L"):
                break
            if not on and needle in name and not name.startswith(".L"):
                on = True
            continue
        if on:
            t = line.strip()
            if t and not t.startswith((";", ".", "//")):
                out.append(t.split()[0])
            if t.startswith("s_endpgm"):
                pass
    return out


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "fma_f64"
    if op.startswith("v_mul_f64"): return "mul_f64"
    if op.startswith("v_add_f64"): return "add_f64"
    if op.startswith(("v_cmp", "v_cmpx")) and "f64" in op: return "cmp_f64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_trig", "v_frexp", "v_ldexp")) : return "div/sqrt pieces"
    if op.startswith(("v_min_f64", "v_max_f64")): return "minmax_f64"
    if op.startswith("v_cndmask"): return "select"
    if op.startswith(("v_mov", "v_accvgpr")): return "move"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "other valu"
    if op.startswith(("s_load", "s_buffer_load")): return "s_load"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    ops = kernel_text(sys.argv[1], sys.argv[2])
    mix = collections.Counter(classify(o) for o in ops)
    valu = sum(v for k, v in mix.items() if k.endswith("f64") or k in ("div/sqrt pieces", "select", "move", "lane", "other valu"))
    print("%s in %s: %d instructions, %d vector ALU" % (sys.argv[2], sys.argv[1], len(ops), valu))
    for k, v in mix.most_common():
        print("  %-18s %6d" % (k, v))


if __name__ == "__main__":
    main()
