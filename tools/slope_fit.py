#!/usr/bin/env python3
"""Measured slope of the bench over database sizes: reads the bench lines given (default profiles/r03*_bench_*.json), fits
ms per batch = a + b x (accelerator records gathered per read) by least squares, and writes profiles/r03_slope.json -- what
bench.py puts into config.extrapolation.fit: the points, the fit and the rate it predicts for the metric's database (31.5 GB
.edx, ~63 G list entries).     python tools/slope_fit.py [files...]"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r03*_bench_*.json")))
pts = []
for f in files:
    for ln in open(f):
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        ex = d["config"]["extrapolation"]
        if d.get("n_gpus", 1) != 1 or "acx_entries_here" not in ex:
            continue
        ph = d["phases_ms_per_batch"]
        pts.append({"file": os.path.basename(f), "edx_gb": ex["this_edx_bytes"] / 1e9, "acx_entries": ex["acx_entries_here"], "records_per_read": ex["acx_records_per_read_here"],
                    "ms_per_batch": d["ms_per_step"], "reads_per_s": d["value"], "ms_prefilter_kernel": ph["ms_prefilter_hash"], "ms_seed_lookups": ph["ms_seed"],
                    "reads_per_batch": d["work"]["entries_per_batch"]})
pts.sort(key=lambda p: p["acx_entries"])
x = np.array([p["records_per_read"] for p in pts]); y = np.array([p["ms_per_batch"] for p in pts]); yp = np.array([p["ms_prefilter_kernel"] for p in pts])
b, a = np.polyfit(x, y, 1)
bp, ap = np.polyfit(x, yp, 1)
rec_per_gentry = float(np.mean([p["records_per_read"] / (p["acx_entries"] / 1e9) for p in pts]))
target_entries = 63e9
xr = rec_per_gentry * target_entries / 1e9
reads = float(np.mean([p["reads_per_batch"] for p in pts]))
out = {"points": pts,
       "fit": {"ms_per_batch": {"a": float(a), "b_per_record_per_read": float(b)}, "ms_prefilter_kernel": {"a": float(ap), "b_per_record_per_read": float(bp)},
               "records_per_read_per_G_entries": rec_per_gentry, "residual_ms": [float(v) for v in (y - (a + b * x))]},
       "metric_database": {"list_entries": target_entries, "records_per_read": xr, "ms_per_batch_predicted": float(a + b * xr), "reads_per_s_predicted": reads / ((a + b * xr) * 1e-3),
                           "note": "one device cannot hold 63 G entries at 5 B each next to the references: the prediction is the per-batch cost of walking that many "
                                   "records per read, i.e. of a device that holds the whole accelerator; with --shard db over S devices each walks 1/S of them"},
       "how": "python tools/slope_fit.py over " + ", ".join(p["file"] for p in pts)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_slope.json"), "w"), indent=1)
for p in pts:
    print("%-28s %6.2f GB .edx %6.2f G entries %6.1f rec/read %5.2f ms/batch (prefilter %4.2f) %6.1f M reads/s" % (p["file"], p["edx_gb"], p["acx_entries"] / 1e9, p["records_per_read"], p["ms_per_batch"], p["ms_prefilter_kernel"], p["reads_per_s"] / 1e6))
print("fit: ms/batch = %.3f + %.5f x records/read;  at 63 G entries: %.0f records/read -> %.2f ms/batch -> %.0f M reads/s" % (a, b, xr, a + b * xr, reads / ((a + b * xr) * 1e-3) / 1e6))
