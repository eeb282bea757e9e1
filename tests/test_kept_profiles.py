"""The measurement set kept under profiles/ for the latest round is ONE set: every bench line, counter file and kernel trace of it names the same
source hash (tools/source_hash.py over the device code), and every bench line either carries its counters or says why not.  No GPU needed: the
files are data."""
import csv
import glob
import json
import os
import re

import pytest

from conftest import ROOT

PROFILES = os.path.join(ROOT, "profiles")


def latest_round():
    rounds = sorted({m.group(1) for f in os.listdir(PROFILES) for m in [re.match(r"(r\d\d)_bench_headline\.json$", f)] if m})
    assert rounds, "profiles/ holds no rNN_bench_headline.json"
    return rounds[-1]


def kept_bench_lines():
    r = latest_round()
    return r, sorted(glob.glob(os.path.join(PROFILES, r + "_bench_*.json")))


def test_kept_bench_lines_are_one_set():
    r, files = kept_bench_lines()
    assert len(files) >= 12, files
    hashes = {}
    for f in files:
        d = json.load(open(f))
        assert d["metric"] and d["unit"] == "Mrays/s" and d["n_gpus"] == 1, f
        hashes[os.path.basename(f)] = d["roofline"]["source_hash"]
    assert len(set(hashes.values())) == 1, hashes
    src = next(iter(hashes.values()))
    for f in glob.glob(os.path.join(PROFILES, r + "_pmc_*.json")) + glob.glob(os.path.join(PROFILES, "pmc_latest*.json")):
        assert json.load(open(f))["source_hash"] == src, "%s is not of the kept set %s" % (f, src)


def test_every_kept_bench_line_has_its_counters_or_says_why_not():
    """a kept line whose launch accounting did not match its counter file used to lose `traffic` and `counters` silently."""
    r, files = kept_bench_lines()
    without = []
    for f in files:
        roof = json.load(open(f))["roofline"]
        if roof.get("counters") is not None:
            assert roof["traffic"] is not None and roof["traffic"] > 0, f
            assert roof["launches_per_step"] > 0 and roof["avg_launch_ms"] > 0, f
            continue
        assert roof.get("counters_note"), "%s has neither counters nor a reason" % f
        without.append(os.path.basename(f))
    # the configurations nobody takes counters for: the 13 s frame and the two batch-budget variants of the headline (their launches are other sizes)
    assert sorted(without) == sorted(r + "_bench_" + n + ".json" for n in ("smallpt_4k_pt1024", "headline_budget_4096mib", "headline_budget_8192mib")), without


def test_kept_kernel_traces_agree_with_the_bench_lines():
    """The dominant kernel's average duration in the kept rocprofv3 trace and the one bench.py measured with HIP events on the kernel's own
    stream belong to the same launches: within 15 % (the trace serialises nothing, the bench overlaps batch lanes)."""
    r, _ = kept_bench_lines()
    for wl, line in (("cornell_pt64", "headline"), ("smallpt_pt64", "smallpt_pt64")):
        stats = os.path.join(PROFILES, "%s_kernel_stats_%s.csv" % (r, wl))
        if not os.path.exists(stats):
            pytest.skip("no kept kernel trace for " + wl)
        roof = json.load(open(os.path.join(PROFILES, "%s_bench_%s.json" % (r, line))))["roofline"]
        # the timed instantiations of the kernel (flag word even; the odd ones are the counting twins of the ray-count pass)
        rows = [row for row in csv.DictReader(open(stats))
                for m in [re.search(r"\b%s<(\d+)," % re.escape(roof["kernel"].split("<")[0]), row["Name"])] if m and int(m.group(1)) % 2 == 0]
        assert rows, (stats, roof["kernel"])
        calls = sum(int(row["Calls"]) for row in rows)
        avg_ms = sum(float(row["TotalDurationNs"]) for row in rows) / calls / 1e6
        assert abs(avg_ms - roof["avg_launch_ms"]) / roof["avg_launch_ms"] < 0.15, (wl, avg_ms, roof["avg_launch_ms"])


def test_kept_headline_carries_both_arithmetics():
    """VERDICT r4 #1a: the headline line reports the opt-in contracted arithmetic beside the exact one -- time, value, parity against the exact frame and
    against the oracle on the sample, and the executed instruction counts of both kernels from the two kept counter files."""
    r, _ = kept_bench_lines()
    d = json.load(open(os.path.join(PROFILES, r + "_bench_headline.json")))
    c = d.get("contracted")
    if c is None:
        pytest.skip("the kept headline predates option fp_contract")
    assert d["config"]["arith"] == "exact" and d["value_contracted"] == c["value"] > d["value"]
    assert c["contracted_launches_per_frame"] > 0 and 1.05 < c["speedup_vs_exact"] < 1.5
    for cmp in (c["vs_exact_frame"], c["vs_oracle_on_the_sample"]):
        assert max(cmp["rms_per_channel"]) <= 1e-4 and cmp["bit_identical_pixels"] >= 0.999
    for k in ("k_pt_bounce", "k_pt_shadow"):
        assert 0.5 < c["valu_wave_instructions_per_launch"][k]["ratio"] < 0.9, k
    # and the exact frame itself is the oracle's on the sample, bit for bit
    assert d["cpu_baseline"]["gpu_frame_vs_oracle_on_the_sample"]["bit_identical_pixels"] == 1.0
