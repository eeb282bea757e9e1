"""Folds the passes of tools/profile_workload.sh into one record per kernel: average launch duration from the
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
    for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"), recursive=True) or glob.glob(os.path.join(out_dir, "kernel_trace.csv")):
        for row in csv.DictReader(open(f, newline="")):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            d = dur.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    kernels = {}
    # every timed variant of a kernel counts (e.g. k_pt_bounce<0, false, false> and the batches' first bounce <0, false, true>): counters and
    # durations are pooled per kernel name, so that the per-launch figures are averages over the launches of a frame, as bench.py's are
    pooled = {}
    for name, v in raw.items():
        k = name.replace("void ", "")
        base = k.split("<")[0]
        if not base.startswith("k_"):
            continue
        if "<" in k:
            flag = k.split("<")[1].split(",")[0].split(">")[0].strip()
            if flag.lstrip("-").isdigit() and (int(flag) & 1):
                continue                  # bit 0 of the flag word = the variant that maintains the work counters: only the others are the timed ones
        g = pooled.setdefault(base, {"counters": {}, "variants": [], "dur": [0, 0.0]})
        g["variants"].append(k)
        for c, e in v.items():
            if isinstance(e, dict) and "total" in e:
                t = g["counters"].setdefault(c, [0.0, 0])
                t[0] += e["total"]; t[1] += e["launches"]
        if k in dur:
            g["dur"][0] += dur[k][0]; g["dur"][1] += dur[k][1]
    for base, g in pooled.items():
        per = {c: t[0] / max(1, t[1]) for c, t in g["counters"].items()}
        rec = {"launches_profiled": max([t[1] for t in g["counters"].values()] or [0]), "per_launch": per, "variants": sorted(g["variants"])}
        if g["dur"][0]:
            rec["avg_launch_ms"] = g["dur"][1] / g["dur"][0]
            rec["launches_traced"] = g["dur"][0]
        k = base
        d = {}
        simd_cycles = per["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 if per.get("GRBM_GUI_ACTIVE") else None     # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
        if "SQ_ACTIVE_INST_VALU" in per and simd_cycles:
            # waves' cycles with a vector instruction in flight, per SIMD cycle: two waves' 32-bit instructions overlap on a SIMD (2 of 4 cycles each),
            # so this reaches 1 only for FP64 code and passes 1 for 32-bit integer code (k_seed) -- an upper estimate of VALU occupancy
            d["valu_wave_active_per_simd_cycle"] = per["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        if "SQ_INSTS_VALU" in per and simd_cycles:
            # issue slots: every vector instruction holds its SIMD for 4 cycles (FP64, and a lone wave's 32-bit) or 2 (32-bit beside other waves): between x2 and x4
            d["valu_issue_x4"] = per["SQ_INSTS_VALU"] * 4.0 / simd_cycles
            d["valu_issue_x2"] = per["SQ_INSTS_VALU"] * 2.0 / simd_cycles
        cu_cycles = per["GRBM_GUI_ACTIVE"] / 8.0 * 256.0 if per.get("GRBM_GUI_ACTIVE") else None          # one texture-address unit and one L1 per CU
        if "TA_TA_BUSY_sum" in per and cu_cycles:
            d["ta_busy"] = per["TA_TA_BUSY_sum"] / cu_cycles
        # where the vector memory path waits, as fractions of the CUs' cycles
        for c, name in (("TCP_TCP_TA_DATA_STALL_CYCLES_sum", "tcp_ta_data_stall"), ("TCP_PENDING_STALL_CYCLES_sum", "tcp_pending_stall"),
                        ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "ta_addr_stalled_by_tc"), ("TA_DATA_STALLED_BY_TC_CYCLES_sum", "ta_data_stalled_by_tc")):
            if c in per and cu_cycles:
                d[name] = per[c] / cu_cycles
        # what one wave-wide instruction costs the L1, reads and writes apart (calibration: tools/ubench/l1_access.hip -- a coalesced dword access is
        # 4 read / 4 write accesses... see profiles/*_l1_access_calibration.txt for the measured table)
        if per.get("TCP_TOTAL_READ_sum") is not None and per.get("TA_FLAT_READ_WAVEFRONTS_sum"):
            d["l1_reads_per_read_instruction"] = per["TCP_TOTAL_READ_sum"] / per["TA_FLAT_READ_WAVEFRONTS_sum"]
        if per.get("TCP_TOTAL_WRITE_sum") is not None and per.get("TA_FLAT_WRITE_WAVEFRONTS_sum"):
            d["l1_writes_per_write_instruction"] = per["TCP_TOTAL_WRITE_sum"] / per["TA_FLAT_WRITE_WAVEFRONTS_sum"]
        if per.get("TCP_TOTAL_READ_sum") and "TCP_TCC_READ_REQ_sum" in per:
            d["l1_read_hit_rate"] = 1.0 - per["TCP_TCC_READ_REQ_sum"] / per["TCP_TOTAL_READ_sum"]
        # the measured instruction mix: wave-instructions per launch by class (tools/pmc_mix.py prints the same from its own passes)
        if per.get("SQ_INSTS_VALU"):
            typed = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]
            if all(("SQ_INSTS_VALU_" + t) in per for t in typed):
                mix = {t.lower(): per["SQ_INSTS_VALU_" + t] for t in typed}
                mix["other"] = per["SQ_INSTS_VALU"] - sum(mix.values())
                mix["fp64_wave_instructions"] = sum(per["SQ_INSTS_VALU_" + t] for t in typed[:4])
                mix["fp64_share_of_valu"] = mix["fp64_wave_instructions"] / per["SQ_INSTS_VALU"]
                rec["instruction_mix_per_launch"] = mix
        if "SQ_THREAD_CYCLES_VALU" in per and per.get("SQ_ACTIVE_INST_VALU"):
            d["lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_ACTIVE_INST_VALU"])
        if per.get("SQ_WAVE_CYCLES"):
            for a, b in (("SQ_WAIT_ANY", "wave_wait_any_frac"), ("SQ_WAIT_INST_ANY", "wave_wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "wave_active_frac")):
                if a in per:
                    d[b] = per[a] / per["SQ_WAVE_CYCLES"]
        if per.get("SQ_WAVE_CYCLES") and simd_cycles:
            # how full the chip was on average over the launch: SQ_WAVE_CYCLES counts resident waves in quad-cycles.  Against the kernel's waves per SIMD
            # (tools/kernel_resources.py) this shows a launch's tail: round 4 found k_whitted at 0.7 of 2-3 (a few tiles set the duration) and the
            # strided wavefront kernels at 3.1-3.3 of 4
            d["mean_resident_waves_per_simd"] = per["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
        if per.get("SQ_WAVES") and "SQ_INSTS_VALU" in per:
            d["valu_insts_per_wave"] = per["SQ_INSTS_VALU"] / per["SQ_WAVES"]
        if "TCC_HIT_sum" in per and (per["TCC_HIT_sum"] + per.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        d["formulas"] = ("valu_wave_active_per_simd_cycle = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (quad-cycles; > 1 possible for 32-bit code); "
                         "valu_issue_xN = SQ_INSTS_VALU x N / SIMD cycles; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wait fractions over SQ_WAVE_CYCLES; "
                         "ta_busy and the four stall fractions = *_sum / (GRBM_GUI_ACTIVE / 8 x 256 CUs); l1_reads_per_read_instruction = TCP_TOTAL_READ_sum / TA_FLAT_READ_WAVEFRONTS_sum, "
                         "writes likewise (calibration: tools/ubench/l1_access.hip); mean_resident_waves_per_simd = SQ_WAVE_CYCLES x 4 / SIMD cycles")
        rec["derived"] = d
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            rec["hbm_bytes_per_launch"] = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
            rec["hbm_bytes_per_launch_fetch_doubled"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
            rec["hbm_note"] = ("rocprofv3 --pmc, separate passes: FETCH_SIZE %.4g KB + WRITE_SIZE %.4g KB per launch, raw.  On gfx950 FETCH_SIZE reads half the bytes of a "
                               "16-B-per-lane stream; this kernel reads 8 and 4 B per lane (uncalibrated width), so the truth lies between raw and *_fetch_doubled"
                               % (per["FETCH_SIZE"], per["WRITE_SIZE"]))
        kernels[base] = rec
    res = {"source_hash": source_hash(), "workload": workload, "mode": "FRAYHIP_PT_LANES=1 (serialised launches)", "kernels": kernels}
    json.dump(res, open(os.path.join(out_dir, "pmc.json"), "w"), indent=1)
    for k, r in sorted(kernels.items()):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r.get("derived", {}).items() if a != "formulas"},
              "avg_launch_ms", r.get("avg_launch_ms"), "hbm_bytes", r.get("hbm_bytes_per_launch"))


if __name__ == "__main__":
    main()
