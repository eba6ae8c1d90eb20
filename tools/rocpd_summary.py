#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) result database as text:
   kernel-trace: per-kernel calls / total / avg / min / max duration;  --pmc: per-kernel average of each counter.
usage: rocpd_summary.py results.db [--pmc] [--short-names]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*?>)?)\(", name)
    if "rocprim" in name:
        k = re.search(r"detail::(\w+)", name.split("trampoline_kernel")[-1][:400])
        return "rocprim::" + (re.findall(r"(radix_sort_\w+|merge_sort_\w+|\w+_kernel)", name) or ["kernel"])[0]
    return m.group(1) if m else name[:60]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    if "--pmc" in sys.argv:
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        print("%-44s %-22s %6s %18s %18s %18s %12s" % ("kernel", "counter", "calls", "avg", "min", "max", "avg_ns"))
        for k, cn, n, a, mn, mx, d in rows:
            print("%-44s %-22s %6d %18.3f %18.3f %18.3f %12.0f" % (short(k)[:44], cn, n, a, mn, mx, d))
    else:
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-44s %6s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
        for k, n, s, a, mn, mx in rows:
            print("%-44s %6d %14d %12.0f %12d %12d %6.2f%%" % (short(k)[:44], n, s, a, mn, mx, 100.0 * s / tot))
        r = c.execute("select vgpr_count, sgpr_count, lds_size, workgroup_x, grid_x, name from kernels group by name").fetchall()
        print("\n%-44s %6s %6s %8s %6s %10s" % ("kernel", "vgpr", "sgpr", "lds", "wg", "grid"))
        for v, s_, l, w, g, k in r:
            print("%-44s %6s %6s %8s %6s %10s" % (short(k)[:44], v, s_, l, w, g))


if __name__ == "__main__":
    main()
