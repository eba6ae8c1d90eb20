#!/usr/bin/env python3
"""Generates the committed golden fixtures from the COMPILED REFERENCE (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the build container only:  python tests/golden/make_golden.py

Outputs (small, data only -- no reference source or binaries):
  refs.fa, q100.fa, q292.fa          seeded synthetic inputs (burst_amd.synth)
  dna.edx, quick.edx                 databases written by the reference (`-d DNA 320 -s 500`, `-d QUICK 320 -s 500`)
  acx.sha256                         sha256 of the .acx files the reference wrote (64 MiB / 4 GiB each, not committed) and of
                                     the QUICK .edx it wrote with -l 0 / -l 40
  *.b6                               sorted reference outputs, one per (db, queries, mode, flags) case; cases.json lists them
  kernel_vectors.npz                 (clump, query, budget) -> MinA, MetaPack from the reference's own kernels
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from burst_amd import synth  # noqa: E402
import dbutil  # noqa: E402
import oraclelib as ol  # noqa: E402

BURST12 = os.path.join(ROOT, "oracle", "_ref", "burst12")
TMP = "/tmp/burst_golden"


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise SystemExit("reference failed: %s" % " ".join(cmd))
    return r.stdout


def make_inputs():
    rng = np.random.default_rng(20240807)
    refs, names = [], []
    for b in range(14):
        base = rng.integers(1, 5, size=int(rng.integers(1500, 2600)), dtype=np.uint8)
        for v, s in enumerate(synth.mutate_family(base, 5, 0.02, rng)):
            if b % 5 == 4:          # a few references carry IUPAC codes / N
                m = np.flatnonzero(rng.random(len(s)) < 0.004)
                s[m] = rng.integers(5, 16, size=len(m))
            refs.append(s)
            names.append("ref%02d_%d some description" % (b, v))
    # exact duplicates (exercise RefDedupIx) and a multi-copy reference (two copies of one region)
    refs.append(refs[3].copy()); names.append("dup_of_ref00_3")
    refs.append(refs[17].copy()); names.append("dup_of_ref03_2")
    rep = np.concatenate([refs[8][:900], rng.integers(1, 5, size=700, dtype=np.uint8), refs[8][200:900]])
    refs.append(rep); names.append("multicopy_ref01")
    refs.append(rng.integers(1, 5, size=90, dtype=np.uint8)); names.append("short_ref")
    synth.write_fasta(os.path.join(HERE, "refs.fa"), refs, names)

    def reads(n, L, edits, seed, iupac, extra):
        r, _ = synth.make_reads(refs, n, L, edits, seed, rc_frac=0.5, iupac_frac=0.0)
        r2, _ = synth.make_reads(refs, n // 6, L, edits[:3], seed + 1, rc_frac=0.5, iupac_frac=iupac)
        r = r + r2
        rngq = np.random.default_rng(seed + 2)
        for i in range(0, len(r), 9):            # duplicated reads (Offset expansion)
            r.append(r[i].copy())
        for i in range(0, len(r), 31):           # reads with N
            x = r[i].copy(); x[int(rngq.integers(0, len(x)))] = 5; r.append(x)
        r += extra
        order = rngq.permutation(len(r))
        r = [r[i] for i in order]
        return r
    extra100 = [refs[0][10:21].copy(), refs[1][5:45].copy(), np.full(100, 1, np.uint8),
                rng.integers(1, 5, size=100, dtype=np.uint8)]
    q100 = reads(420, 100, [0, 1, 2, 3, 4, 6], 42, 0.02, extra100)
    synth.write_fasta(os.path.join(HERE, "q100.fa"), q100, prefix="q")
    q292 = reads(120, 292, [0, 2, 5, 9, 12], 43, 0.01, [])
    synth.write_fasta(os.path.join(HERE, "q292.fa"), q292, prefix="amp")


def sorted_b6(path_in, path_out):
    with open(path_in, "rb") as f:
        lines = sorted(f.read().splitlines())
    with open(path_out, "wb") as f:
        for ln in lines:
            f.write(ln + b"\n")
    return len(lines)


def make_b6():
    os.makedirs(TMP, exist_ok=True)
    refs = os.path.join(HERE, "refs.fa")
    shas = {}
    for kind in ("DNA", "QUICK"):
        edx = os.path.join(HERE, "%s.edx" % kind.lower())
        acx = os.path.join(TMP, "%s.acx" % kind.lower())
        run([BURST12, "-r", refs, "-d", kind, "320", "-o", edx, "-a", acx, "-s", "500", "-i", "0.95", "-t", "1"])
        shas[kind.lower() + ".acx"] = hashlib.sha256(open(acx, "rb").read()).hexdigest()
    # an accelerator built with -y (N as wildcard) for the -y case; the .edx it writes is identical to dna.edx
    edx_y, acx_y = os.path.join(TMP, "dna_y.edx"), os.path.join(TMP, "dna_y.acx")
    run([BURST12, "-r", refs, "-d", "DNA", "320", "-o", edx_y, "-a", acx_y, "-s", "500", "-i", "0.95", "-t", "1", "-y"])
    assert open(edx_y, "rb").read() == open(os.path.join(HERE, "dna.edx"), "rb").read()
    shas["dna_y.acx"] = hashlib.sha256(open(acx_y, "rb").read()).hexdigest()
    # DB15 (15-mers, 4 GiB length table): the QUICK database again with the reference compiled with -DSCOUR_N=15
    edx15, acx15 = os.path.join(TMP, "quick15.edx"), os.path.join(TMP, "quick15.acx")
    run([BURST12[:-2] + "15", "-r", refs, "-d", "QUICK", "320", "-o", edx15, "-a", acx15, "-s", "500", "-i", "0.95", "-t", "1"])
    assert open(edx15, "rb").read() == open(os.path.join(HERE, "quick.edx"), "rb").read()
    h15 = hashlib.sha256()
    with open(acx15, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h15.update(blk)
    shas["quick_k15.acx"] = h15.hexdigest()
    os.remove(acx15)
    # clump formation tolerance -l (LATENCY): 0 = input order, 40 = wider pods than the default 16
    for lat in (0, 40):
        e = os.path.join(TMP, "quick_l%d.edx" % lat)
        run([BURST12, "-r", refs, "-d", "QUICK", "320", "-o", e, "-s", "500", "-i", "0.95", "-t", "1", "-l", str(lat)])
        shas["quick_l%d.edx" % lat] = hashlib.sha256(open(e, "rb").read()).hexdigest()
    json.dump(shas, open(os.path.join(HERE, "acx.sha256"), "w"), indent=1)
    cases = []

    def case(name, db, q, mode, ident, extra=(), accel=True, threads=4):
        out = os.path.join(TMP, name + ".raw")
        cmd = [BURST12, "-q", os.path.join(HERE, q), "-o", out, "-m", mode, "-i", ident, "-t", str(threads), "--noprogress"]
        if db == "fasta":
            cmd += ["-r", refs]
        else:
            cmd += ["-r", os.path.join(HERE, db + ".edx")]
            if accel:
                cmd += ["-a", os.path.join(TMP, db + ("_y" if "-y" in extra else "") + ".acx")]
        cmd += list(extra)
        run(cmd)
        n = sorted_b6(out, os.path.join(HERE, name + ".b6"))
        cases.append({"name": name, "db": db, "queries": q, "mode": mode, "id": ident, "extra": list(extra), "accel": accel, "threads": threads, "lines": n})
        print("%-32s %6d lines" % (name, n))

    for mode in ("BEST", "ALLPATHS", "CAPITALIST", "FORAGE"):
        case("dna_q100_%s_fr" % mode.lower(), "dna", "q100.fa", mode, "0.95", ["-fr"])
    case("dna_q100_best", "dna", "q100.fa", "BEST", "0.97")
    case("dna_q100_allpaths_y", "dna", "q100.fa", "ALLPATHS", "0.95", ["-fr", "-y"])
    case("dna_q100_allpaths_noacx_fr", "dna", "q100.fa", "ALLPATHS", "0.95", ["-fr"], accel=False)
    case("quick_q100_capitalist_fr", "quick", "q100.fa", "CAPITALIST", "0.97", ["-fr"])
    case("quick_q100_forage", "quick", "q100.fa", "FORAGE", "0.96")
    case("dna_q292_allpaths_fr", "dna", "q292.fa", "ALLPATHS", "0.97", ["-fr"])
    case("dna_q292_forage_fr", "dna", "q292.fa", "FORAGE", "0.95", ["-fr"])
    case("quick_q292_best_fr", "quick", "q292.fa", "BEST", "0.96", ["-fr"])
    # the reference's deterministic configuration (one thread, exhaustive path): hit-list order is reproducible there,
    # so the order-sensitive modes (DUPE_HUNT / CAPITALIST ties) can be pinned exactly
    case("dna_q100_capitalist_noacx_t1_fr", "dna", "q100.fa", "CAPITALIST", "0.95", ["-fr"], accel=False, threads=1)
    case("dna_q100_forage_noacx_t1_fr", "dna", "q100.fa", "FORAGE", "0.95", ["-fr"], accel=False, threads=1)
    case("dna_q292_forage_noacx_t1_fr", "dna", "q292.fa", "FORAGE", "0.95", ["-fr"], accel=False, threads=1)
    case("quick_q100_capitalist_noacx_t1", "quick", "q100.fa", "CAPITALIST", "0.97", [], accel=False, threads=1)
    case("fasta_q100_best", "fasta", "q100.fa", "BEST", "0.97", ["-s"])
    case("fasta_q100_allpaths_fr", "fasta", "q100.fa", "ALLPATHS", "0.95", ["-fr", "-s"])
    case("fasta_q100_best_noshear", "fasta", "q100.fa", "BEST", "0.97")
    json.dump(cases, open(os.path.join(HERE, "cases.json"), "w"), indent=1)


def make_taxonomy():
    """taxonomy map for refs.fa: 7-level strings shared family-wise, a few shallower ones, one empty, some headers absent;
    keys with and without the description so that either form of the stored reference header finds its line"""
    names = [ln[1:].rstrip("\n") for ln in open(os.path.join(HERE, "refs.fa")) if ln.startswith(">")]
    lines = []
    for nm in names:
        key = nm.split()[0]
        if key.startswith("ref"):
            b, v = int(key[3:5]), int(key.split("_")[1])
            if b == 11:
                continue                                   # no taxonomy for this family
            tax = "k__Bacteria;p__P%d;c__C%d;o__O%d;f__F%d;g__G%d;s__S%d_%d" % (b % 2, b % 3, b % 5, b // 2, b, b, v % 3)
            if b == 7:
                tax = ";".join(tax.split(";")[:4 + v % 3])  # shallower assignments
            if b == 9 and v == 1:
                tax = ""                                    # empty taxonomy string
        elif key.startswith("dup_of_ref00"):
            tax = "k__Bacteria;p__P0;c__C0;o__O0;f__F0;g__G0;s__S0_0"
        elif key.startswith("multicopy"):
            tax = "k__Bacteria;p__P1;c__C1;o__O1;f__F0;g__G1"
        else:
            continue
        lines.append("%s\t%s" % (key, tax))
        if nm != key:
            lines.append("%s\t%s" % (nm, tax))
    with open(os.path.join(HERE, "tax.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def make_tax_cases():
    """column 13: the reference with -b / -bs / -bc on the committed inputs; appended to cases.json"""
    os.makedirs(TMP, exist_ok=True)
    make_taxonomy()
    tax = os.path.join(HERE, "tax.txt")
    refs = os.path.join(HERE, "refs.fa")
    acx = os.path.join(TMP, "dna.acx")
    if not os.path.exists(acx):
        run([BURST12, "-r", refs, "-d", "DNA", "320", "-o", os.path.join(TMP, "dna_again.edx"), "-a", acx, "-s", "500", "-i", "0.95", "-t", "1"])
    cases = [c for c in json.load(open(os.path.join(HERE, "cases.json"))) if "-b" not in c["extra"]]

    def case(name, db, q, mode, ident, extra=(), accel=True, threads=4):
        out = os.path.join(TMP, name + ".raw")
        cmd = [BURST12, "-q", os.path.join(HERE, q), "-o", out, "-m", mode, "-i", ident, "-t", str(threads), "--noprogress", "-r", os.path.join(HERE, db + ".edx")]
        if accel:
            cmd += ["-a", acx]
        cmd += [tax if a == "tax.txt" else a for a in extra]
        run(cmd)
        n = sorted_b6(out, os.path.join(HERE, name + ".b6"))
        cases.append({"name": name, "db": db, "queries": q, "mode": mode, "id": ident, "extra": list(extra), "accel": accel, "threads": threads, "lines": n})
        print("%-40s %6d lines" % (name, n))
    case("dna_q100_capitalist_tax_noacx_t1_fr", "dna", "q100.fa", "CAPITALIST", "0.95", ["-fr", "-b", "tax.txt"], accel=False, threads=1)
    case("dna_q100_capitalist_tax_bs_noacx_t1_fr", "dna", "q100.fa", "CAPITALIST", "0.95", ["-fr", "-b", "tax.txt", "-bs"], accel=False, threads=1)
    case("dna_q100_capitalist_tax_bc3_noacx_t1_fr", "dna", "q100.fa", "CAPITALIST", "0.95", ["-fr", "-b", "tax.txt", "-bc", "3"], accel=False, threads=1)
    case("dna_q292_capitalist_tax_bs_strict_noacx_t1_fr", "dna", "q292.fa", "CAPITALIST", "0.95", ["-fr", "-b", "tax.txt", "-bs", "STRICT"], accel=False, threads=1)
    case("dna_q100_best_tax_fr", "dna", "q100.fa", "BEST", "0.95", ["-fr", "-b", "tax.txt"])
    case("dna_q100_best_tax_bs_strict_fr", "dna", "q100.fa", "BEST", "0.95", ["-fr", "-b", "tax.txt", "-bs", "STRICT"])
    case("dna_q100_allpaths_tax_fr", "dna", "q100.fa", "ALLPATHS", "0.95", ["-fr", "-b", "tax.txt"])
    case("dna_q100_forage_tax_noacx_t1_fr", "dna", "q100.fa", "FORAGE", "0.95", ["-fr", "-b", "tax.txt"], accel=False, threads=1)
    json.dump(cases, open(os.path.join(HERE, "cases.json"), "w"), indent=1)


def make_kernel_vectors():
    rng = np.random.default_rng(7)
    R = ol.reference()
    R.ref_setscore(1)
    rec = {k: [] for k in ("rows", "rows_off", "q", "q_off", "E", "ret", "mins", "score_bits", "finalPos", "gapR", "gapQ", "ret_forage",
                           "score_bits_f", "finalPos_f", "gapR_f", "gapQ_f", "mins_f")}
    rows_off, q_off = [0], [0]
    for case_i in range(160):
        qlen = int(rng.choice([24, 50, 100, 100, 100, 150, 292, 320]))
        E = int(rng.choice([1, 2, 3, 5, 9, 16]))
        E = min(E, max(1, qlen // 8))
        L = int(rng.integers(qlen + 10, qlen + 400))
        base = rng.integers(1, 5, size=L, dtype=np.uint8)
        if case_i % 7 == 0:      # low-complexity stretch
            st = int(rng.integers(0, L - 50)); base[st:st + 50] = np.resize(rng.integers(1, 5, size=int(rng.integers(1, 4)), dtype=np.uint8), 50)
        refs = []
        for z in range(16):
            v = synth.apply_edits(base, int(rng.integers(0, 7)), rng)
            if z % 5 == 0:
                v = v[:int(rng.integers(max(12, qlen // 2), len(v)))]
            if case_i % 4 == 0:
                m = np.flatnonzero(rng.random(len(v)) < 0.01); v[m] = rng.integers(5, 16, size=len(m))
            refs.append(v)
        src = refs[int(rng.integers(0, 16))]
        st = int(rng.integers(0, max(1, len(src) - qlen)))
        q = synth.apply_edits(src[st:st + qlen], int(rng.integers(0, E + 3)), rng)
        if case_i % 6 == 0:
            m = np.flatnonzero(rng.random(len(q)) < 0.02); q[m] = rng.integers(5, 16, size=len(m))
        if q[0] == 0 or len(q) < 2:
            continue
        rows = dbutil.clump_rows(refs, 0)
        ret, mins, score, fin, gr, gq = ol.ref_align_clump(rows, q, E, variant=0)
        ret1, mins1, *_ = ol.ref_align_clump(rows, q, E, variant=1)
        assert ret == ret1 and np.array_equal(mins, mins1)
        retf, minsf, scoref, finf, grf, gqf = ol.ref_align_clump(rows, q, E, variant=0, bound_override=E)
        rec["rows"].append(rows.reshape(-1)); rows_off.append(rows_off[-1] + rows.size)
        rec["q"].append(q); q_off.append(q_off[-1] + len(q))
        rec["E"].append(E); rec["ret"].append(ret); rec["mins"].append(mins)
        rec["score_bits"].append(score.view(np.uint32)); rec["finalPos"].append(fin); rec["gapR"].append(gr); rec["gapQ"].append(gq)
        rec["ret_forage"].append(retf); rec["mins_f"].append(minsf)
        rec["score_bits_f"].append(scoref.view(np.uint32)); rec["finalPos_f"].append(finf); rec["gapR_f"].append(grf); rec["gapQ_f"].append(gqf)
    np.savez_compressed(os.path.join(HERE, "kernel_vectors.npz"),
                        rows=np.concatenate(rec["rows"]), rows_off=np.array(rows_off, np.int64),
                        q=np.concatenate(rec["q"]), q_off=np.array(q_off, np.int64), E=np.array(rec["E"], np.uint32),
                        ret=np.array(rec["ret"], np.uint32), mins=np.stack(rec["mins"]),
                        score_bits=np.stack(rec["score_bits"]), finalPos=np.stack(rec["finalPos"]), gapR=np.stack(rec["gapR"]), gapQ=np.stack(rec["gapQ"]),
                        ret_forage=np.array(rec["ret_forage"], np.uint32), mins_f=np.stack(rec["mins_f"]), score_bits_f=np.stack(rec["score_bits_f"]),
                        finalPos_f=np.stack(rec["finalPos_f"]), gapR_f=np.stack(rec["gapR_f"]), gapQ_f=np.stack(rec["gapQ_f"]))
    print("kernel vectors:", len(rec["E"]), "cases,", int(sum((m != 255).sum() for m in rec["mins"])), "hit lanes")


if __name__ == "__main__":
    if not os.path.exists(BURST12):
        raise SystemExit("build the reference first: make -C oracle ref")
    if len(sys.argv) > 1 and sys.argv[1] == "tax":      # only (re)generate the taxonomy cases
        make_tax_cases()
        raise SystemExit(0)
    make_inputs()
    make_b6()
    make_tax_cases()
    make_kernel_vectors()
