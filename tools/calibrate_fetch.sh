#!/bin/bash
# FETCH_SIZE on the prefilter's access pattern: tools/ubench/run_gather (runs of N consecutive dwords at random 4-byte-aligned addresses,
# 64 adjacent lanes on 64 adjacent positions) under rocprofv3 --pmc FETCH_SIZE, its LAST dispatch against the bytes the program knows it
# touched at 32 / 64 / 128-byte granularity.  -> gpurun_out/<tag>_fetch_calibration_runs.txt (copy into profiles/).
#   gpurun -- 'bash tools/calibrate_fetch.sh r06x'
cd /tmp && export TMPDIR=/tmp
R=/root/repo; TAG=${1:-cal}; O=$R/gpurun_out; mkdir -p $O
X=$R/tools/ubench/run_gather
[ -x $X ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/run_gather.hip -o $X || exit 1
OUT=$O/${TAG}_fetch_calibration_runs.txt; : > $OUT
for RUN in 40 16 164 1; do
	rm -rf $O/${TAG}_cal
	rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cal -- $X 12 $RUN 64 8 > $O/${TAG}_cal.log 2>&1
	python - "$O/${TAG}_cal" "$O/${TAG}_cal.log" "$RUN" >> $OUT <<'PY'
import csv, glob, sys, re
d, log, run = sys.argv[1:4]
txt = open(log).read()
m = re.search(r"useful_bytes (\d+)\s+bytes_as_32B_pieces (\d+)\s+bytes_as_64B_pieces (\d+)\s+bytes_as_128B_pieces (\d+)", txt)
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k_runs" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print("runs of %s dwords: %s" % (run, [l for l in txt.splitlines() if l.startswith("rate")][0] if "rate" in txt else txt[-200:]))
if m and rows:
    fetch = float(rows[-1]["Counter_Value"]) * 1024.0
    u, b32, b64, b128 = map(float, m.groups())
    print("  FETCH_SIZE x 1024 = %.4g B (last dispatch) | useful %.4g  32-B pieces %.4g  64-B pieces %.4g  128-B pieces %.4g" % (fetch, u, b32, b64, b128))
    print("  FETCH_SIZE / useful = %.3f   / 32-B = %.3f   / 64-B = %.3f   / 128-B = %.3f" % (fetch / u, fetch / b32, fetch / b64, fetch / b128))
else:
    print("  no counter rows (%d) or no program output" % len(rows))
PY
done
rm -rf $O/${TAG}_cal; cat $OUT
