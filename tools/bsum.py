"""one-line summary of bench.py JSON lines read from stdin: python bench.py ... | python tools/bsum.py [label]"""
import sys, json
label = sys.argv[1] if len(sys.argv) > 1 else ""
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    w = d["work"]
    print(label, "%.2f Mreads/s %.2f ms" % (d["value"] / 1e6, d["ms_per_step"]), {k[3:]: round(v, 2) for k, v in d["phases_ms_per_batch"].items()},
          "tasks/read %.2f" % w.get("lane_tasks_per_read", 0), "ent/read %.0f" % w["acx_entries_per_read"], "hits", w["hits"], d["roofline"]["kernel"])
