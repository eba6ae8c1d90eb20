#!/usr/bin/env python3
"""Adds the `-m ANY` cases to the committed golden fixtures (cases.json + one sorted .b6 each) from the COMPILED REFERENCE
(oracle/_ref/burst12), using the inputs make_golden.py committed.  Run in the build container only:
    python tests/golden/make_golden_any.py
ANY prints the first hit within budget a thread meets (burst.c:4239-4275, 4457-4475; column 12 = duplicate flag).  With one thread
and no accelerator that is deterministic (clumps ascending, entries in sorted order, lanes ascending) and pinned exactly; with the
accelerator the visiting order follows the bunch k-mer counts and the thread count, so those cases carry the relaxed contract of
tests/goldenlib.py (same queries, same duplicate flags, every line a placement the device path computes)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BURST12 = os.path.join(ROOT, "oracle", "_ref", "burst12")
TMP = "/tmp/burst_golden_any"


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise SystemExit("reference failed: %s" % " ".join(cmd))


def main():
    os.makedirs(TMP, exist_ok=True)
    refs = os.path.join(HERE, "refs.fa")
    acx = {}
    for kind in ("DNA", "QUICK"):
        edx = os.path.join(TMP, "%s.edx" % kind.lower())
        acx[kind.lower()] = os.path.join(TMP, "%s.acx" % kind.lower())
        run([BURST12, "-r", refs, "-d", kind, "320", "-o", edx, "-a", acx[kind.lower()], "-s", "500", "-i", "0.95", "-t", "1"])
        assert open(edx, "rb").read() == open(os.path.join(HERE, "%s.edx" % kind.lower()), "rb").read(), "the reference no longer writes the committed .edx"
    cases = [c for c in json.load(open(os.path.join(HERE, "cases.json"))) if c["mode"] != "ANY"]

    def case(name, db, q, ident, extra=(), accel=True, threads=4):
        out = os.path.join(TMP, name + ".raw")
        cmd = [BURST12, "-q", os.path.join(HERE, q), "-o", out, "-m", "ANY", "-i", ident, "-t", str(threads), "--noprogress", "-r", os.path.join(HERE, db + ".edx")]
        if accel:
            cmd += ["-a", acx[db]]
        run(cmd + list(extra))
        lines = sorted(open(out, "rb").read().splitlines())
        with open(os.path.join(HERE, name + ".b6"), "wb") as f:
            for ln in lines:
                f.write(ln + b"\n")
        cases.append({"name": name, "db": db, "queries": q, "mode": "ANY", "id": ident, "extra": list(extra), "accel": accel, "threads": threads, "lines": len(lines)})
        print("%-36s %6d lines" % (name, len(lines)))

    case("dna_q100_any_noacx_t1_fr", "dna", "q100.fa", "0.95", ["-fr"], accel=False, threads=1)
    case("quick_q100_any_noacx_t1", "quick", "q100.fa", "0.96", [], accel=False, threads=1)
    case("dna_q292_any_noacx_t1_fr", "dna", "q292.fa", "0.95", ["-fr"], accel=False, threads=1)
    case("dna_q100_any_fr", "dna", "q100.fa", "0.95", ["-fr"])
    case("quick_q100_any", "quick", "q100.fa", "0.97", [])
    json.dump(cases, open(os.path.join(HERE, "cases.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
