"""Host logic without a GPU: the C host library on both sides of the ORACLE (tests/csrc/cpu_e2e.c) must reproduce
the reference's .b6 for the golden cases -- parsers, query pipeline, .edx reader, direct-FASTA clumping,
per-mode consolidation, coordinates and formatting.  A subset of the cases keeps the CPU suite short; the GPU
suite (test_gpu_e2e.py) runs all of them through the real device path."""
import os
import subprocess

import pytest

import goldenlib as gl

ROOT = gl.ROOT
EXE = "/tmp/burst_amd_cpu_e2e"
SUBSET = ["dna_q100_best_fr", "dna_q100_allpaths_y", "dna_q100_capitalist_noacx_t1_fr", "dna_q292_forage_noacx_t1_fr",
          "quick_q100_capitalist_fr", "fasta_q100_allpaths_fr", "fasta_q100_best_noshear",
          # column 13 (taxonomy): lookup, CAPITALIST interpolation, -bc, -bs / -bs STRICT
          # (the other taxonomy cases run through the command line in the gpu suite)
          "dna_q100_capitalist_tax_noacx_t1_fr", "dna_q100_capitalist_tax_bs_noacx_t1_fr", "dna_q100_capitalist_tax_bc3_noacx_t1_fr",
          "dna_q100_best_tax_bs_strict_fr", "dna_q100_forage_tax_noacx_t1_fr",
          # ANY = the first hit within budget the reference's single thread meets (exact), column 12 = duplicate flag
          "dna_q100_any_noacx_t1_fr", "quick_q100_any_noacx_t1", "dna_q292_any_noacx_t1_fr", "dna_q100_any_fr", "quick_q100_any"]


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fopenmp", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "burst_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "csrc", "cpu_e2e.c"), "-o", EXE,
                           "-L" + os.path.join(ROOT, "burst_amd"), "-lburst_host", "-lburst_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "burst_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
    return EXE


@pytest.mark.parametrize("name", SUBSET)
def test_host_pipeline_matches_reference(exe, name, tmp_path):
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")

    def run(flags, mode=None):
        tax, bs, strict, cut = gl.tax_args(c)
        subprocess.check_call([exe, ref, q, out, mode or c["mode"], c["id"], str(fr), str(z), "0" if shear else "-1", str(flags), tax, str(bs), str(strict), str(cut)])
        return sorted(open(out, "rb").read().splitlines())
    base = 0 if c["accel"] else 1
    got = run(base)
    # (the placements an order-dependent line may be: everything the mode computes, without the duplicate hunt; ANY: what FORAGE computes)
    nd = run(base | 2, "FORAGE" if c["mode"] == "ANY" else None) if gl.order_sensitive(c) else None
    gl.compare(c, got, nd)


@pytest.mark.parametrize("name", ["dna_q100_capitalist_tax_noacx_t1_fr", "dna_q292_forage_noacx_t1_fr", "dna_q100_allpaths_y"])
def test_threaded_report_equals_sequential(exe, name, tmp_path, monkeypatch):
    """the consolidation renders chunks of queries on several threads and writes them in order: forced on with small chunks"""
    monkeypatch.setenv("BURST_HOST_REPORT_THREADS", "5:37")
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    tax, bs, strict, cut = gl.tax_args(c)
    subprocess.check_call([exe, ref, q, out, c["mode"], c["id"], str(fr), str(z), "0" if shear else "-1", "0" if c["accel"] else "1", tax, str(bs), str(strict), str(cut)])
    threaded = open(out, "rb").read()
    monkeypatch.setenv("BURST_HOST_REPORT_THREADS", "1")
    subprocess.check_call([exe, ref, q, out, c["mode"], c["id"], str(fr), str(z), "0" if shear else "-1", "0" if c["accel"] else "1", tax, str(bs), str(strict), str(cut)])
    assert threaded == open(out, "rb").read()                 # same bytes in the same order
    gl.compare(c, sorted(threaded.splitlines()), None if not gl.order_sensitive(c) else sorted(threaded.splitlines()))


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def test_database_builders_write_the_reference_bytes(tmp_path):
    """`-d QUICK` + shear (burst.c:1840-1858, 2109-2190, 2687-2741), dump_edb (2758-2839) and
    make_accelerator (3304-3532).  The .edx must be byte-identical to the one the reference wrote for the same FASTA and
    the .acx must have the sha256 of the reference's; --make-acx from the finished .edx must give the same accelerator.
    (No device is involved in database construction.)"""
    import json
    cli = os.path.join(ROOT, "burst_amd", "burst_hip")
    want = json.load(open(os.path.join(gl.G, "acx.sha256")))
    edx, acx, acx2 = str(tmp_path / "q.edx"), str(tmp_path / "q.acx"), str(tmp_path / "q2.acx")
    subprocess.check_call([cli, "-r", os.path.join(gl.G, "refs.fa"), "-d", "QUICK", "320", "-o", edx, "-a", acx, "-s", "500", "-i", "0.95"],
                          stdout=subprocess.DEVNULL)
    assert open(edx, "rb").read() == open(os.path.join(gl.G, "quick.edx"), "rb").read()
    assert _sha256(acx) == want["quick.acx"]
    subprocess.check_call([cli, "-r", os.path.join(gl.G, "quick.edx"), "--make-acx", acx2], stdout=subprocess.DEVNULL)
    assert _sha256(acx2) == want["quick.acx"]
    for lat in (0, 40):                # clump formation tolerance -l (burst.c:83, 2149-2189)
        subprocess.check_call([cli, "-r", os.path.join(gl.G, "refs.fa"), "-d", "QUICK", "320", "-o", edx, "-s", "500", "-i", "0.95", "-l", str(lat)],
                              stdout=subprocess.DEVNULL)
        assert _sha256(edx) == want["quick_l%d.edx" % lat]


def test_database_builders_differential(tmp_path):
    """tools/db_diff.py: `-d QUICK` of burst_hip next to the compiled reference on random and awkward reference FASTA files
    (duplicate sequences and fragments, sequences shorter than K, IUPAC codes, N runs, lower case, CRLF, wrapped lines, no
    final newline, 1 and 17 sequences) x five parameter sets (shear lengths, query lengths, identities, -y, -l 0): .edx and
    .acx byte-identical.  The order of identical fragments is whatever the reference's sort calls leave behind; the same
    calls are made (bh_db.c, "Clump formation order")."""
    import sys
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst12")):
        pytest.skip("compiled reference not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "db_diff.py"), "2", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-4000:]


def test_fastq_and_gzip_queries_load_like_fasta(tmp_path):
    """bh_queries_load reads gzip (magic 1f 8b) and FASTQ ('@' records of four lines) besides the reference's two-line FASTA
    (burst.c:636-690; the reference itself reads neither): the query tables must come out identical."""
    import gzip
    import numpy as np
    from burst_amd import host
    src = os.path.join(gl.G, "q100.fa")
    recs = open(src).read().split("\n")
    recs = [r for r in recs if r != ""]
    fq = tmp_path / "q.fastq"
    with open(fq, "w") as f:
        for h, s in zip(recs[0::2], recs[1::2]):
            f.write("@%s\n%s\n+\n%s\n" % (h[1:], s, "I" * len(s)))
    fagz, fqgz = tmp_path / "q.fa.gz", tmp_path / "q.fastq.gz"
    with gzip.open(fagz, "wb") as g:
        g.write(open(src, "rb").read())
    with gzip.open(fqgz, "wb") as g:
        g.write(open(fq, "rb").read())

    def tables(path):
        qs = host.QuerySet(str(path), 0.95, rc=True, accel=True, K=12)
        c = qs.c
        off = host._view(c.qoff, qs.n_entries + 1, np.uint64).copy()
        t = (qs.n_reads, qs.n_uniq, off.tobytes(), host._view(c.codes, int(off[-1]), np.uint8).tobytes(), host._view(c.emac, qs.n_entries, np.uint16).tobytes(),
             host._view(c.offset, qs.n_uniq + 1, np.uint64).tobytes(), host._view(c.flags, qs.n_entries, np.uint8).tobytes(),
             host._view(c.codes4, (int(off[-1]) + 1) // 2, np.uint8).tobytes())
        qs.close()
        return t
    base = tables(src)
    assert base[0] == len(recs) // 2 and base[1] > 0
    # blank lines between records and at the end of the file, CRLF line ends
    fq2 = tmp_path / "q_blank.fastq"
    with open(fq2, "w", newline="") as f:
        for k, (h, s) in enumerate(zip(recs[0::2], recs[1::2])):
            f.write("@%s\n%s\n+%s\n%s\n" % (h[1:], s, h[1:] if k % 3 == 0 else "", "I" * len(s)))
            if k % 5 == 0:
                f.write("\n")
        f.write("\n\n")
    for p in (fq, fagz, fqgz, fq2):
        assert tables(p) == base, p
    # wrapped records, a missing separator and a truncated last record are refused with a message, not read as garbage
    for name, text in (("wrapped", "@r1\nACGT\nACGT\n+\nIIII\nIIII\n"), ("nosep", "@r1\nACGT\nIIII\n@r2\nACGT\n+\nIIII\n"), ("short", "@r1\nACGT\n+\nIIII\n@r2\nACGT\n")):
        bad = tmp_path / (name + ".fastq")
        bad.write_text(text)
        with pytest.raises(host.HostError) as e:
            host.QuerySet(str(bad), 0.95, rc=False, accel=True, K=12)
        assert "FASTQ" in str(e.value)
    # the other packed copies: four symbols per byte for A/C/G/T (code - 1), 2-byte lengths, clean-batch counts
    qs = host.QuerySet(str(src), 0.95, rc=True, accel=True, K=12)
    off = host._view(qs.c.qoff, qs.n_entries + 1, np.uint64)
    codes_all = host._view(qs.c.codes, int(off[-1]), np.uint8)
    c2 = host._view(qs.c.codes2, (int(off[-1]) + 3) // 4, np.uint8)
    pad4 = np.concatenate([codes_all, np.ones(4, np.uint8)])
    two = ((pad4 - 1) & 3).astype(np.uint8)
    n2 = len(c2)
    want2 = two[0::4][:n2] | (two[1::4][:n2] << 2) | (two[2::4][:n2] << 4) | (two[3::4][:n2] << 6)
    assert np.array_equal(c2[:-1], want2[:-1])
    assert np.array_equal(host._view(qs.c.len16, qs.n_uniq, np.uint16), np.diff(off[:qs.n_uniq + 1]).astype(np.uint16))
    amb = host._view(qs.c.ambBefore, qs.n_uniq + 1, np.uint32)
    per = np.array([int(((codes_all[int(off[i]):int(off[i + 1])] - 1) > 3).any()) for i in range(qs.n_uniq)])
    assert amb[0] == 0 and np.array_equal(np.diff(amb.astype(np.int64)), per) and per.sum() > 0
    qs.close()
    # packed copy: two symbols per byte, low nibble first
    codes = np.frombuffer(base[3], np.uint8)
    c4 = np.frombuffer(base[7], np.uint8)
    pad = np.concatenate([codes, np.zeros(1, np.uint8)])
    assert np.array_equal(c4, (pad[0:len(c4) * 2:2] & 15) | ((pad[1:len(c4) * 2:2] & 15) << 4))


def test_report_identity_column_equals_printf():
    """the report formats its lines by hand; column 3 must be what the reference's fprintf("%f") prints for the f32 product
    score * 100 (burst.c:4553-4557) -- compared for every identity 1 - ed / (len + gapQ) with len up to 1 100, for the products'
    float neighbours, and for values outside the usual range (which take the snprintf path)"""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(ROOT, "burst_amd", "libburst_host.so"))
    lib.bh_report_format_identities.restype = C.c_uint64
    lib.bh_report_format_identities.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
    ln = np.arange(1, 1101, dtype=np.float32)[None, :, None]
    ed = np.arange(0, 255, dtype=np.float32)[:, None, None]
    gq = np.arange(0, 4, dtype=np.float32)[None, None, :]
    score = (np.float32(1.0) - ed / (ln + gq)).astype(np.float32).reshape(-1)
    score = np.unique(score)
    extra = np.array([0.0, 1.0, 0.5, 1e-7, 0.999999, 0.9999995, 123456.789, -0.25, 3.0e7, np.inf], np.float32)
    vals = np.concatenate([score, np.nextafter(score, np.float32(2)), np.nextafter(score, np.float32(-2)), extra]).astype(np.float32)
    buf = C.create_string_buffer(len(vals) * 48)
    n = lib.bh_report_format_identities(vals.ctypes.data, len(vals), buf, len(buf))
    got = buf.raw[:n].decode().split("\n")[:-1]
    pct = (vals * np.float32(100)).astype(np.float32)
    want = ["%f" % float(v) for v in pct]
    assert len(got) == len(want) > 500000
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]


def test_database_built_in_parts_is_the_same_set_of_references(tmp_path):
    """bh_edx_merge: a database built part by part (bench.py, databases too large to be sorted in one piece in the memory at hand) and
    laid end to end holds the same fragments as the one-piece build, and every reference number still resolves to the header and the
    start of the sequence its lane was cut from"""
    import ctypes as C
    import sys
    import types
    import numpy as np
    sys.path.insert(0, gl.ROOT)
    import bench
    from burst_amd import host
    a = types.SimpleNamespace(read_len=100, n_base=5000, n_variants=2, ref_len=900, variant_rate=0.05, id=0.98, K=15, edits="0,1,2", reads=500, pool=4, iupac=0.0, fr=False, drop_refs=False)
    old = bench.PART_BASES
    try:
        bench.PART_BASES = 2000          # three parts
        _, edx3, _, reads3, _ = bench.build_inputs(str(tmp_path / "parts"), types.SimpleNamespace(**vars(a)), 0)
        bench.PART_BASES = old
        refs1, edx1, _, _, _ = bench.build_inputs(str(tmp_path / "one"), types.SimpleNamespace(**vars(a)), 0)
    finally:
        bench.PART_BASES = old
    names = [ln[1:].strip() for ln in open(reads3) if ln.startswith(">")]
    assert len(names) == 2000 and len({n.split("_")[0] for n in names}) == 2000          # unique read names over the parts
    fasta = {}
    with open(refs1) as f:
        for h in f:
            fasta[h[1:].strip().encode()] = f.readline().strip().encode()
    code = {65: 1, 67: 2, 71: 3, 84: 4}

    def lanes(path):
        db = host.Db.read(path)
        cl = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
        pk = host._view(db.c.packed, db.c.packedWords * 16, np.uint8)
        srt = host._view(db.c.refIxSrt, db.c.totR, np.uint32)
        start = host._view(db.c.refStart, db.c.origTotR, np.uint32)
        heads = C.cast(db.c.refHead, C.POINTER(C.c_char_p))
        out, w = [], 0
        for c in range(db.c.numRclumps):
            L = int(cl[c]); rows = (L + 1) // 2
            blk = pk[w * 16:(w + rows) * 16].reshape(rows, 16); w += rows
            full = np.empty((rows * 2, 16), np.uint8)
            full[0::2] = blk & 15; full[1::2] = blk >> 4
            for z in range(16):
                i = 16 * c + z
                if i < db.c.totR:
                    seq = bytes(full[:L, z]).rstrip(b"\0")
                    out.append((seq, heads[int(srt[i])], int(start[int(srt[i])])))
        tot = (db.c.totR, db.c.origTotR, db.c.numRclumps, db.c.numRefHeads)
        db.close()
        return out, tot
    l3, t3 = lanes(edx3)
    l1, t1 = lanes(edx1)
    assert t3 == t1 and sorted(x[0] for x in l3) == sorted(x[0] for x in l1)
    for seq, head, st in l3[::7]:
        src = fasta[head][st:st + len(seq)]
        assert bytes(code[b] for b in src) == seq, (head, st)


def test_database_parts_with_duplicate_fragments_and_partial_clumps(tmp_path):
    """bh_edx_merge on parts that carry duplicate-fragment tables (bench.py --db-profile strains: families of near-identical sequences
    shear into identical fragments) and do not fill their last clump: the merged file's own RefDedupIx holds the parts' tables, the
    identity for parts without one and EMPTY ranges for the padding lanes in the middle.  Every unique reference of the merged database
    resolves to exactly the originals (header, start) whose sequence it is, every original belongs to one unique reference, and the
    compiled reference reads the file and reports on it what it reports on the parts' union of sequences."""
    import ctypes as C
    import subprocess
    import sys
    import types
    import numpy as np
    sys.path.insert(0, gl.ROOT)
    import bench
    from burst_amd import host
    a = types.SimpleNamespace(read_len=100, n_base=3000, n_variants=2, ref_len=900, variant_rate=0.05, id=0.98, K=12, edits="0,1,2", reads=150, pool=2, iupac=0.0, fr=False,
                              drop_refs=False, db_profile="strains")
    old = bench.PART_BASES
    try:
        bench.PART_BASES = 1000
        specs = bench.db_parts(a)
        assert len(specs) >= 5 and {nv for _, _, nv, _, _ in specs} == {2, 60, 200, 500}
        _, edx, _, reads, _ = bench.build_inputs(str(tmp_path / "strains"), a, 0)
    finally:
        bench.PART_BASES = old
    db = host.Db.read(edx)
    totR, orig = int(db.c.totR), int(db.c.origTotR)
    assert totR < orig and db.c.refDedupIx                     # duplicates exist and are folded
    dd = host._view(db.c.refDedupIx, totR + 1, np.uint32).astype(np.int64)
    rix = host._view(db.c.tmpRIX, orig, np.uint32)
    start = host._view(db.c.refStart, orig, np.uint32)
    cl = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    pk = host._view(db.c.packed, db.c.packedWords * 16, np.uint8)
    heads = C.cast(db.c.refHead, C.POINTER(C.c_char_p))
    assert dd[0] == 0 and dd[-1] == orig and (np.diff(dd) >= 0).all()
    assert sorted(rix.tolist()) == list(range(orig))          # every original fragment belongs to exactly one unique reference
    n_pad = int((np.diff(dd) == 0).sum())
    assert 0 < n_pad < 16 * len(specs)                          # padding lanes: at most one partial clump per part
    # regenerate the sequences the parts were built from (a sequence depends on the seed and its number only) and check a sample of lanes
    code = {65: 1, 67: 2, 71: 3, 84: 4}
    fasta = {}
    for p, (b0, nb, nv, rate, seed) in enumerate(specs):
        fa = str(tmp_path / ("p%d.fa" % p))
        host.synth_refs(fa, nb, nv, a.ref_len, rate, seed, first_base=b0)
        with open(fa) as f:
            for hline in f:
                fasta[hline[1:].strip().encode()] = f.readline().strip().encode()
    w, checked = 0, 0
    for c in range(db.c.numRclumps):
        L = int(cl[c]); rows = (L + 1) // 2
        if c % 7 == 0:
            blk = pk[w * 16:(w + rows) * 16].reshape(rows, 16)
            full = np.empty((rows * 2, 16), np.uint8)
            full[0::2] = blk & 15; full[1::2] = blk >> 4
            for z in range(16):
                i = 16 * c + z
                if i >= totR:
                    continue
                seq = bytes(full[:L, z]).rstrip(b"\0")
                if dd[i] == dd[i + 1]:
                    assert seq == b""                          # a padding lane: nothing but pads
                    continue
                for k in range(int(dd[i]), int(dd[i + 1])):
                    o = int(rix[k])
                    src = fasta[heads[o]][int(start[o]):int(start[o]) + len(seq)]
                    assert bytes(code[b] for b in src) == seq, (i, o)
                    checked += 1
        w += rows
    assert checked > 1000
    db.close()
    exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12")
    if os.path.exists(exe):          # the compiled reference accepts the merged file and finds the reads in it
        out = str(tmp_path / "ref.b6")
        r = subprocess.run([exe, "-r", edx, "-q", reads, "-o", out, "-m", "BEST", "-i", "0.98", "-t", "4", "--noprogress"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-500:]
        lines = open(out).read().splitlines()
        assert len(lines) >= 0.97 * 300                             # (reads of 97 symbols with two edits are beyond the budget)


@pytest.mark.parametrize("no_pty", [False, True])
def test_reference_align_phase_is_stamped_with_and_without_a_pty(tmp_path, monkeypatch, no_pty):
    """bench.py's cpu_baseline times the reference's alignment loops between two of its own progress lines; the GPU box has no pty
    devices (`out of pty devices`), so the lines must also arrive one by one through `stdbuf -oL` on a pipe"""
    import shutil
    import sys
    sys.path.insert(0, gl.ROOT)
    import bench
    exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12")
    if not os.path.exists(exe):
        pytest.skip("compiled reference not built")
    if no_pty:
        if not shutil.which("stdbuf"):
            pytest.skip("no stdbuf")
        import pty

        def refuse():
            raise OSError("out of pty devices")
        monkeypatch.setattr(pty, "openpty", refuse)
    out = str(tmp_path / "o.b6")
    rc, wall, align, tail = bench.run_reference_timed([exe, "-r", os.path.join(gl.G, "dna.edx"), "-q", os.path.join(gl.G, "q100.fa"), "-o", out, "-m", "BEST", "-i", "0.97", "-t", "2"])
    assert rc == 0, tail
    assert align is not None and 0 < align <= wall
    assert sorted(open(out)) == sorted(open(os.path.join(gl.G, "dna_q100_best.b6")))


def test_host_pipeline_with_reads_beyond_1024_symbols(exe, tmp_path):
    """reads of 1 100 .. 3 900 symbols (and a few of 100) against unsheared 4 500-symbol references searched directly: the C host on both
    sides of the oracle against the compiled reference run here (-t 1) -- ingest, budgets, consolidation and coordinates for queries the
    rounds before refused; the device side of the same thing is tests/test_gpu_e2e.py::test_command_line_with_reads_beyond_1024_symbols"""
    import numpy as np
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "burst12")
    if not os.path.exists(ref_exe):
        pytest.skip("compiled reference not built")
    rng = np.random.default_rng(5)
    A = np.array(list("ACGT"))

    def mutate(x, n):
        x = list(x)
        for _ in range(n):
            k = int(rng.integers(3)); i = int(rng.integers(1, len(x) - 1))
            if k == 0:
                x[i] = "ACGT"[("ACGT".index(x[i]) + 1 + int(rng.integers(3))) % 4]
            elif k == 1:
                del x[i]
            else:
                x.insert(i, "ACGT"[int(rng.integers(4))])
        return "".join(x)
    refs = []
    for f in range(2):
        base = "".join(A[rng.integers(0, 4, size=4500)])
        refs += [("fam%d_v%d" % (f, v), mutate(base, 30 * v)) for v in range(4)]
    refs_fa, q_fa = str(tmp_path / "refs.fa"), str(tmp_path / "q.fa")
    open(refs_fa, "w").write("".join(">%s\n%s\n" % r for r in refs))
    reads = []
    for i in range(12):
        h, s = refs[int(rng.integers(len(refs)))]
        n = 100 if i % 4 == 3 else int(rng.integers(1100, 3901))
        st = int(rng.integers(0, len(s) - n))
        reads.append(("r%d_%s" % (i, h), mutate(s[st:st + n], int(rng.integers(0, max(2, n // 60))))))
    open(q_fa, "w").write("".join(">%s\n%s\n" % r for r in reads))
    for mode, ident, fr in (("BEST", "0.97", 0), ("ALLPATHS", "0.96", 1), ("FORAGE", "0.95", 0)):
        want, got = str(tmp_path / "ref.b6"), str(tmp_path / "hip.b6")
        subprocess.check_call([ref_exe, "-r", refs_fa, "-q", q_fa, "-o", want, "-m", mode, "-i", ident, "-t", "1", "--noprogress"] + (["-fr"] if fr else []),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([exe, refs_fa, q_fa, got, mode, ident, str(fr), "1", "-1", "1", "", "0", "0", "10"])
        a, b = sorted(open(want, "rb").read().splitlines()), sorted(open(got, "rb").read().splitlines())
        assert len(a) >= 12 and a == b, (mode, len(a), len(b), sorted(set(a) ^ set(b))[:4])


@pytest.mark.parametrize("alphabet,mode", [("ACDEFGHIKLMN", "BEST"), ("ACDEFGHIKLMN", "ALLPATHS"), ("ACGTacgtN", "ALLPATHS"), ("01", "FORAGE"), ("ACDEFGHIKLMNPQR", "CAPITALIST")])
def test_xalphabet_matches_the_reference(exe, tmp_path, alphabet, mode):
    """-x ("any alphabet, unambiguous ID matching": aded_xalpha / reScoreM_xalpha, burst.c:696-697, 894, 1099): raw symbols compared for
    equality, against FASTA references, forward strand.  The device layout has four bits per symbol, so the run's alphabet (up to 15
    distinct bytes) is mapped onto codes 1..15 in byte order and scored by the identity table.  Upstream, -x only survives symbols whose
    BYTE VALUE is below 16: its query sort buckets the first five raw bytes as nibbles (NIB5, burst.c:383-387) and writes out of bounds
    for any printable letter (segmentation fault in "Sorting queries...").  The compiled reference therefore gets the same sequences
    with the symbols renamed to the bytes 1, 2, ... (skipping 10 and 13, the line ends) in the same order -- a .b6 holds no sequence, so
    the two outputs must be identical: protein-like letters, case-sensitive nucleotides (a != A), a binary alphabet, fifteen symbols."""
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "burst12")
    if not os.path.exists(ref_exe):
        pytest.skip("compiled reference not present")
    import xalpha_util
    files = xalpha_util.write_inputs(tmp_path, alphabet, len(alphabet) * 7 + len(mode))
    if files is None:
        pytest.skip("fifteen symbols need a byte beyond 15 once the line ends are left out: the reference cannot run this one")
    rf, qf, rf_low, qf_low = files
    out_ref, out_mine = str(tmp_path / "ref.b6"), str(tmp_path / "mine.b6")
    r = subprocess.run([ref_exe, "-x", "-r", rf_low, "-q", qf_low, "-o", out_ref, "-m", mode, "-i", "0.93", "-t", "1", "--noprogress"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-800:]
    subprocess.check_call([exe, rf, qf, out_mine, mode, "0.93", "0", "1", "-1", "1", "", "0", "0", "10"], env=dict(os.environ, BURST_XALPHA="1"))
    a, b = sorted(open(out_ref, "rb").read().splitlines()), sorted(open(out_mine, "rb").read().splitlines())
    assert len(a) > 100 and a == b


def test_accelerator_window_walk_eight_symbols_at_a_time(tmp_path):
    """burst_amd/csrc/bhip_acx_words.h (acx_lane_words: the scans of the device's accelerator builders) on the host: where a dword of
    symbols is all A/C/G/T with K - 1 such symbols in front, the eight windows are shifts of one register of packed 2-bit codes; anywhere
    else the symbol-by-symbol walk of make_accelerator (burst.c:3343-3377) with IUPAC expansion.  Same multiset of words as that walk
    alone on random lanes -- ambiguity codes at 0 / 0.4 / 3 / 20 %, padding inside and junk behind the lane, K = 4 .. 15, both N rules."""
    exe = str(tmp_path / "acx_words_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "burst_amd", "csrc"), os.path.join(ROOT, "tests", "csrc", "acx_words_host.cpp"), "-o", exe])
    for seed in (5, 11):
        r = subprocess.run([exe, "12000", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
