"""Instruction mix per kernel from tools/pmc_mix.sh's raw sums (tools/pmc_summarise.py): wave-instructions per launch by class, their share of
SQ_INSTS_VALU, and what is left over (moves, selects, compares, bit operations ... : everything the typed counters do not name).

usage: python tools/pmc_mix.py RAW.json [--json]"""
import json
import sys

TYPED = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]


def mix(raw):
    out = {}
    for k, v in raw.items():
        if "SQ_INSTS_VALU" not in v:
            continue
        n = max(1, v["SQ_INSTS_VALU"]["launches"])
        per = lambda c: (v[c]["total"] / max(1, v[c]["launches"])) if c in v else None      # noqa: E731
        valu = per("SQ_INSTS_VALU")
        if not valu:
            continue
        e = {"launches_profiled": n, "valu": valu, "salu": per("SQ_INSTS_SALU"), "smem": per("SQ_INSTS_SMEM"), "waves": per("SQ_WAVES"),
             "vmem_rd": per("SQ_INSTS_VMEM_RD"), "vmem_wr": per("SQ_INSTS_VMEM_WR"), "lds": per("SQ_INSTS_LDS"), "branch": per("SQ_INSTS_BRANCH")}
        typed = 0.0
        for t in TYPED:
            x = per("SQ_INSTS_VALU_" + t)
            e[t.lower()] = x
            typed += x or 0.0
        e["other_valu"] = valu - typed
        e["fp64_share"] = sum(e[t.lower()] or 0.0 for t in TYPED[:4]) / valu
        tc = per("SQ_THREAD_CYCLES_VALU")
        av = per("SQ_ACTIVE_INST_VALU")
        if tc and av:
            e["lanes_active"] = tc / (av * 64.0) if av else None
        out[k] = e
    return out


def main():
    raw = json.load(open(sys.argv[1]))
    m = mix(raw)
    if "--json" in sys.argv:
        print(json.dumps(m, indent=1))
        return
    for k, e in sorted(m.items(), key=lambda kv: -kv[1]["valu"] * kv[1]["launches_profiled"]):
        if e["valu"] < 1e5:
            continue
        print("%s   (%d launches)" % (k, e["launches_profiled"]))
        print("   VALU %.4g per launch; SALU %.4g, SMEM %.4g, VMEM rd/wr %s/%s, LDS %s, branch %s, waves %s" % (
            e["valu"], e["salu"] or 0, e["smem"] or 0, e["vmem_rd"], e["vmem_wr"], e["lds"], e["branch"], e["waves"]))
        print("   " + "  ".join("%s %.1f%%" % (t.lower(), 100.0 * (e[t.lower()] or 0) / e["valu"]) for t in TYPED) + "  other %.1f%%" % (100.0 * e["other_valu"] / e["valu"]))


if __name__ == "__main__":
    main()
