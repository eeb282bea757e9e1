#!/bin/bash
# Calibration of the L1 per-instruction figures (run on the GPU box from the repo root): tools/ubench/l1_access.hip under rocprofv3 --pmc.
#   tools/ubench/run_l1_access.sh OUTDIR
set -e
OUT=$1; mkdir -p $OUT; OUT=$(cd $OUT && pwd); ROOT=$(pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/ubench/l1_access.hip -o /tmp/l1_access
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/l1p
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT/l1p -- /tmp/l1_access > $OUT/l1_access.log 2>&1
cd $ROOT
python3 - $OUT/l1p > $OUT/l1_access_calibration.txt <<'PY'
import csv, glob, os, sys
raw = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"]
        k = k[:k.rindex("(")] if k.endswith(")") and "(" in k else k          # the argument list only: "k_load<unsigned int __vector(2), 8>" keeps its own name
        e = raw.setdefault(k, {}).setdefault(row["Counter_Name"], {"total": 0.0})
        e["total"] += float(row["Counter_Value"])
waves, iters = 2048 * 4, 64
print("kernel: L1 reads / writes per wave-wide instruction (TCP_TOTAL_READ_sum, TCP_TOTAL_WRITE_sum over waves x instructions); TA wavefront counters per instruction")
for k, v in raw.items():
    if "k_load" not in k and "k_store" not in k:
        continue
    per = lambda c: v[c]["total"] / (waves * iters) if c in v else float("nan")
    print("%-34s reads %7.2f  writes %7.2f  accesses %7.2f   ta_read_wavefronts %.3f  ta_write_wavefronts %.3f" % (
        k.replace("void ", ""), per("TCP_TOTAL_READ_sum"), per("TCP_TOTAL_WRITE_sum"), per("TCP_TOTAL_ACCESSES_sum"), per("TA_FLAT_READ_WAVEFRONTS_sum"), per("TA_FLAT_WRITE_WAVEFRONTS_sum")))
PY
cat $OUT/l1_access_calibration.txt
rm -rf $OUT/l1p
