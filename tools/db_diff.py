#!/usr/bin/env python3
"""Differential run of the two database builders: `burst_hip -d QUICK` (accelerator on the device when there is one -- DB_DIFF_EXPECT_DEVICE=1
insists on it --, else the host builder) vs the compiled reference
(oracle/_ref/burst12) on random and awkward reference FASTA files -- the .edx files must be byte-identical and the .acx
files too.
   python tools/db_diff.py [n_random] [workdir]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "burst12")
CLI = os.path.join(ROOT, "burst_amd", "burst_hip")
n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 12
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/db_diff"
os.makedirs(work, exist_ok=True)
rng = np.random.default_rng(11)
ACGT = np.frombuffer(b"ACGT", np.uint8)
IUPAC = np.frombuffer(b"RYKMSWBDHVN", np.uint8)


def rand_seq(n, iupac=0.0, lower=False):
    s = ACGT[rng.integers(0, 4, size=n)].copy()
    if iupac:
        m = rng.random(n) < iupac
        s[m] = IUPAC[rng.integers(0, len(IUPAC), size=int(m.sum()))]
    s = s.tobytes().decode()
    return s.lower() if lower else s


def write(path, recs, eol="\n", wrap=0, final_nl=True):
    with open(path, "w", newline="") as f:
        for i, (h, s) in enumerate(recs):
            f.write(">" + h + eol)
            last = i == len(recs) - 1
            if wrap and s:
                chunks = [s[k:k + wrap] for k in range(0, len(s), wrap)]
                f.write(eol.join(chunks) + ("" if last and not final_nl else eol))
            else:
                f.write(s + ("" if last and not final_nl else eol))


def family(n_base, n_var, length, rate, **kw):
    out = []
    for b in range(n_base):
        base = rand_seq(int(length * rng.uniform(0.6, 1.4)), **kw)
        for v in range(n_var):
            s = list(base)
            for p in np.flatnonzero(rng.random(len(s)) < rate):
                s[p] = "ACGT"[int(rng.integers(4))]
            if rng.random() < 0.3:
                cut = int(rng.integers(0, max(1, len(s) // 10)))
                s = s[cut:]
            out.append(("r%d_%d desc %d" % (b, v, v), "".join(s)))
    return out


cases = []
for i in range(n_random):
    recs = family(int(rng.integers(1, 6)), int(rng.integers(1, 12)), int(rng.integers(60, 2500)), float(rng.choice([0.0, 0.01, 0.05])),
                  iupac=float(rng.choice([0, 0, 0.002, 0.02])))
    if i % 3 == 0:
        recs += [recs[0], (recs[-1][0] + "_dup", recs[-1][1])]                  # exact duplicates
    if i % 4 == 1:
        recs += [("tiny%d" % k, rand_seq(int(rng.integers(1, 14)))) for k in range(3)]   # shorter than K
    order = rng.permutation(len(recs))
    recs = [recs[k] for k in order]
    p = os.path.join(work, "rand%d.fa" % i)
    write(p, recs, wrap=int(rng.choice([0, 0, 60, 70])))
    cases.append(("rand%d" % i, p))
awk = family(3, 5, 900, 0.02)
write(os.path.join(work, "lower.fa"), [(h, s.lower()) for h, s in awk]); cases.append(("lowercase", os.path.join(work, "lower.fa")))
write(os.path.join(work, "crlf.fa"), awk, eol="\r\n"); cases.append(("crlf", os.path.join(work, "crlf.fa")))
write(os.path.join(work, "nonl.fa"), awk, final_nl=False); cases.append(("no_final_newline", os.path.join(work, "nonl.fa")))
write(os.path.join(work, "one.fa"), awk[:1]); cases.append(("single_sequence", os.path.join(work, "one.fa")))
write(os.path.join(work, "seventeen.fa"), family(1, 17, 300, 0.01)); cases.append(("seventeen", os.path.join(work, "seventeen.fa")))
write(os.path.join(work, "ambig.fa"), family(2, 4, 700, 0.01, iupac=0.08)); cases.append(("heavy_iupac", os.path.join(work, "ambig.fa")))
write(os.path.join(work, "nrun.fa"), [(h, s[:200] + "N" * 60 + s[260:]) for h, s in awk]); cases.append(("n_runs", os.path.join(work, "nrun.fa")))

params = [["-d", "QUICK", "100", "-s", "500", "-i", "0.97"], ["-d", "QUICK", "320", "-s", "-i", "0.95"], ["-d", "QUICK", "150", "-i", "0.98"],
          ["-d", "QUICK", "100", "-s", "200", "-i", "0.9", "-y"], ["-d", "QUICK", "250", "-s", "1000", "-i", "0.97", "-l", "0"],
          ["-d", "QUICK", "120", "-s", "400", "-i", "0.96", "-sa"]]
# (case, parameters) pairs side by side, a few at a time: each is two short processes; the lines are printed in the order of the loops
import concurrent.futures


def one_pair(job):
    k, name, fa, par = job
    outs = []
    for exe, tail, tag in ((REF, ["-t", "1"], "ref"), (CLI, [], "hip")):
        edx, acx = os.path.join(work, "%s_%d.edx" % (tag, k)), os.path.join(work, "%s_%d.acx" % (tag, k))
        for f in (edx, acx):
            if os.path.exists(f):
                os.remove(f)
        r = subprocess.run([exe, "-r", fa, "-o", edx, "-a", acx] + par + tail, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        outs.append((r.returncode, open(edx, "rb").read() if os.path.exists(edx) else None, open(acx, "rb").read() if os.path.exists(acx) else None, r.stdout[-400:]))
        for f in (edx, acx):
            if os.path.exists(f):
                os.remove(f)
    same = outs[0][0] == outs[1][0] and (outs[0][0] != 0 or (outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]))
    if os.environ.get("DB_DIFF_EXPECT_DEVICE") == "1" and outs[1][0] == 0 and "-sa" not in par and "built on device" not in outs[1][3]:
        same = False
    text = ["%-18s %-44s ref rc=%d hip rc=%d edx %s acx %s  %s" % (name, " ".join(par), outs[0][0], outs[1][0],
            "same" if outs[0][1] == outs[1][1] else "DIFFERENT", "same" if outs[0][2] == outs[1][2] else "DIFFERENT", "ok" if same else "DIFF")]
    if not same:
        text += ["   ref: " + outs[0][3].replace("\n", " | ")[-300:], "   hip: " + outs[1][3].replace("\n", " | ")[-300:]]
    return (0 if same else 1), text


bad = 0
jobs = [(k, name, fa, par) for k, (name, fa, par) in enumerate((name, fa, par) for name, fa in cases for par in params)]
with concurrent.futures.ThreadPoolExecutor(max_workers=int(os.environ.get("DB_DIFF_JOBS", "6"))) as pool:
    for b_, text in pool.map(one_pair, jobs):
        bad += b_
        print("\n".join(text), flush=True)
print("db_diff:", "ALL OK" if not bad else "%d differing runs" % bad)
sys.exit(1 if bad else 0)
