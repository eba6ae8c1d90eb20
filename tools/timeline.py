"""kernel timeline of one steady-state batch from a rocprofv3 kernel trace: python tools/timeline.py <dir with *_kernel_trace.csv> [batches from the end]
(start, end, duration in microseconds relative to the batch's prefilter kernel -- its seed lookups run during the batch before --; queue id; kernel)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def first_pass(n):      # the prefilter kernel that opens a batch's chain (not the second pass over overflowed queries: k_prefilter_cq<M, 1>, k_prefilter_cf<11, 4>)
    return ("k_prefilter_cq<" in n and ", 0>" in n) or ("k_prefilter_cw<" in n and ", 0>" in n) or ("k_prefilter_cf<" in n and "<11, 4>" not in n) or "k_prefilter_mask" in n
idx = [i for i, r in enumerate(rows) if first_pass(r["Kernel_Name"])]
if len(idx) < back + 1:
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_seed_ranges")]
i0, i1 = idx[-back], idx[-back + 1]
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = None
for r in rows[i0 - 3:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f  q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "")[:70]))
print("batch period: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
