"""Folds the passes of tools/profile_headline.sh into one record per kernel: average launch duration from the
kernel trace, HBM traffic from FETCH_SIZE / WRITE_SIZE, VALU issue / lane utilisation / wait fractions from the
SQ counters, tagged with the source hash of the device code they were measured on.

usage: python tools/pmc_finish.py gpurun_out/prof_TAG WORKLOAD   ->  gpurun_out/prof_TAG/pmc.json
Copy that file to profiles/<round>_pmc.json and profiles/pmc_latest.json to have bench.py print it."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.source_hash import source_hash  # noqa: E402


def main():
    out_dir, workload = sys.argv[1], sys.argv[2]
    raw = json.load(open(os.path.join(out_dir, "pmc_raw.json")))
    # kernel trace of the serialised run: average duration per kernel
    dur = {}
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            d = dur.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    kernels = {}
    for name, v in raw.items():
        k = name.replace("void ", "")
        base = k.split("<")[0]
        if not base.startswith("k_"):
            continue
        if "<" in k and k.split("<")[1].split(",")[0].split(">")[0].strip() != "0":
            continue                      # only the uninstrumented variants (flag word 0) are the timed ones
        per = {c: e["total"] / max(1, e["launches"]) for c, e in v.items() if isinstance(e, dict) and "total" in e}
        rec = {"launches_profiled": max([e["launches"] for c, e in v.items() if isinstance(e, dict) and "launches" in e] or [0]), "per_launch": per}
        if k in dur:
            rec["avg_launch_ms"] = dur[k][1] / dur[k][0]
            rec["launches_traced"] = dur[k][0]
        d = {}
        if "SQ_ACTIVE_INST_VALU" in per and per.get("GRBM_GUI_ACTIVE"):
            d["valu_busy"] = per["SQ_ACTIVE_INST_VALU"] * 4.0 / (per["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_THREAD_CYCLES_VALU" in per and per.get("SQ_ACTIVE_INST_VALU"):
            d["lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_ACTIVE_INST_VALU"])
        if per.get("SQ_WAVE_CYCLES"):
            for a, b in (("SQ_WAIT_ANY", "wave_wait_any_frac"), ("SQ_WAIT_INST_ANY", "wave_wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "wave_active_frac")):
                if a in per:
                    d[b] = per[a] / per["SQ_WAVE_CYCLES"]
        if per.get("SQ_WAVES") and "SQ_INSTS_VALU" in per:
            d["valu_insts_per_wave"] = per["SQ_INSTS_VALU"] / per["SQ_WAVES"]
        if "TCC_HIT_sum" in per and (per["TCC_HIT_sum"] + per.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        d["formulas"] = ("valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                         "wait fractions over SQ_WAVE_CYCLES")
        rec["derived"] = d
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            rec["hbm_bytes_per_launch"] = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
            rec["hbm_bytes_per_launch_fetch_doubled"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
            rec["hbm_note"] = ("rocprofv3 --pmc, separate passes: FETCH_SIZE %.4g KB + WRITE_SIZE %.4g KB per launch, raw.  On gfx950 FETCH_SIZE reads half the bytes of a "
                               "16-B-per-lane stream; this kernel reads 8 and 4 B per lane (uncalibrated width), so the truth lies between raw and *_fetch_doubled"
                               % (per["FETCH_SIZE"], per["WRITE_SIZE"]))
        rec["variant"] = k
        prev = kernels.get(base)
        if prev is None or rec["launches_profiled"] > prev["launches_profiled"]:      # e.g. k_pt_bounce<0, false> and <0, true>: the one that ran
            kernels[base] = rec
    res = {"source_hash": source_hash(), "workload": workload, "mode": "FRAYHIP_PT_LANES=1 (serialised launches)", "kernels": kernels}
    json.dump(res, open(os.path.join(out_dir, "pmc.json"), "w"), indent=1)
    for k, r in sorted(kernels.items()):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r.get("derived", {}).items() if a != "formulas"},
              "avg_launch_ms", r.get("avg_launch_ms"), "hbm_bytes", r.get("hbm_bytes_per_launch"))


if __name__ == "__main__":
    main()
