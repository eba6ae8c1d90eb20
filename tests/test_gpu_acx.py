"""The accelerator built ON THE DEVICE from the references alone (bhip_init with K and no tables: make_accelerator,
burst.c:3304-3532, as gfx950 kernels -- word tuples of every lane, IUPAC expansion, the reference's BadList budget, radix sort,
fold) against the tables of the .acx path: the host builder's (itself byte-identical to the compiled reference's files,
tests/test_host_cpu.py) loaded through read_accelerator's layout, and against the compiled reference directly."""
import os
import subprocess
import sys

import numpy as np
import pytest

import goldenlib as gl

pytestmark = pytest.mark.gpu
CLI = os.path.join(gl.ROOT, "burst_amd", "burst_hip")


_FILE_SIDE = {}          # (path, K, z) -> the tables of the file path, kept while one test builds the same database several ways


def _compare_built_with_loaded(db_path_or_db, K, z, from_fasta=None, files=True):
    from burst_amd import host
    L = host.lib()
    import ctypes as C
    # tables of the file path: host builder -> upload -> export (once per database / K / z: at K = 15 the table of list lengths alone is
    # 4.3 GB, and a test that builds the same database four ways was rebuilding and re-exporting it four times)
    key = (from_fasta or db_path_or_db, K, z)
    if from_fasta:
        b = host.Db.from_fasta(from_fasta, 120, 0.95, shear_len=500)
    else:
        b = host.Db.read(db_path_or_db)
    if key not in _FILE_SIDE:
        if from_fasta:
            a = host.Db.from_fasta(from_fasta, 120, 0.95, shear_len=500, K=K, z=z)
        else:
            a = host.Db.read(db_path_or_db)
            host._chk(L.bh_acx_build(C.byref(a.c), K, z))
        dev_a = a.open_device(0, z)
        lens_a, clumps_a, masks_a, bad_a = dev_a.acx_export(K)
        dev_a.close()
        # what went up is what the host builder made
        assert np.array_equal(lens_a, host._view(a.c.acxLens, 1 << (2 * K), np.uint32))
        assert np.array_equal(bad_a, host._view(a.c.badList, a.c.badSz, np.uint32))
        _FILE_SIDE.clear()
        _FILE_SIDE[key] = (lens_a, clumps_a, masks_a, bad_a, int(a.c.acxFmt), host._view(a.c.acxLists, a.c.acxListBytes, np.uint8).copy())
        a.close()
    lens_a, clumps_a, masks_a, bad_a, fmt_a, lists_a = _FILE_SIDE[key]
    dev_b = b.open_device(0, z, build_K=K)
    lens_b, clumps_b, masks_b, bad_b = dev_b.acx_export(K)
    assert np.array_equal(lens_a, lens_b)
    assert np.array_equal(clumps_a, clumps_b)
    assert np.array_equal(bad_a, bad_b)
    # lane masks: the built ones are exact per lane; the loaded path adds every lane of a clump that holds IUPAC symbols
    # (and every lane for words it knows from the expansion only): a superset, equal where no ambiguity is involved
    assert not np.any(masks_b & ~masks_a)
    assert np.all(masks_b != 0)
    # the .acx the host writes from the device's tables is the host builder's, byte for byte
    b.acx_from_device(dev_b, K, z)
    assert b.c.acxFmt == fmt_a and b.c.acxListBytes == len(lists_a)
    assert np.array_equal(host._view(b.c.acxLists, b.c.acxListBytes, np.uint8), lists_a)
    # ... and so is the file streamed from the device run by run (bh_acx_write_from_device: what bench.py writes for the reference)
    # (files=False: a repeated build of the same database through another builder path -- the files, 4.3 GB each at K = 15, were compared by the first)
    if files:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            f1, f2 = os.path.join(td, "held.acx"), os.path.join(td, "streamed.acx")
            host._chk(L.bh_acx_write(C.byref(b.c), f1.encode()))
            b.acx_write_from_device(dev_b, K, f2, z)
            assert open(f1, "rb").read() == open(f2, "rb").read()
    dev_b.close()
    n = len(clumps_a)
    b.close()
    return n, len(bad_a), int(np.count_nonzero(masks_a != masks_b))


@pytest.mark.parametrize("db,K,z", [("dna", 12, 1), ("dna", 12, 0), ("quick", 12, 1), ("quick", 10, 0), ("quick", 15, 1)])
def test_device_built_accelerator_equals_file(db, K, z, monkeypatch):
    """golden databases (references with IUPAC codes and N): same list lengths, same clump ids in the same order, same BadList"""
    n, nbad, _ = _compare_built_with_loaded(os.path.join(gl.G, db + ".edx"), K, z)      # (the default builder: slices of the word space)
    assert n > 10000
    monkeypatch.setenv("BHIP_ACX_BUILD", "clumps")                                         # the clump-sliced builder, everything at once
    n1, _, _ = _compare_built_with_loaded(os.path.join(gl.G, db + ".edx"), K, z, files=False)
    assert n1 == n
    if K == 12:      # both in small slices (words: several scans; clumps: two passes -- list lengths, then records at running list positions)
        for how in ("clumps", "words"):
            monkeypatch.setenv("BHIP_ACX_BUILD", how)
            monkeypatch.setenv("BHIP_MASK_SLICE", "40000")
            n2, _, _ = _compare_built_with_loaded(os.path.join(gl.G, db + ".edx"), K, z, files=False)
            assert n2 == n
            monkeypatch.delenv("BHIP_MASK_SLICE")
    monkeypatch.delenv("BHIP_ACX_BUILD")
    # ... and through the path of the large databases (BHIP_TEST_TWO_PLANS: a counting pass over the sorted tuples, the record area one
    # address range whose memory is mapped by a thread beside that pass and cut back to the real size, a second plan for the records)
    monkeypatch.setenv("BHIP_TEST_TWO_PLANS", "1")
    n3, _, _ = _compare_built_with_loaded(os.path.join(gl.G, db + ".edx"), K, z, files=False)
    assert n3 == n
    monkeypatch.setenv("BHIP_ACX_NO_PREMAP", "1")          # (the same with the record area allocated in one piece after the counting pass)
    n4, _, _ = _compare_built_with_loaded(os.path.join(gl.G, db + ".edx"), K, z, files=False)
    assert n4 == n
    _FILE_SIDE.clear()


def test_device_built_accelerator_expansion_and_badlist(tmp_path, monkeypatch):
    """heavy ambiguity: runs of three-way codes (hundreds of thousands of expanded words per clump), N runs, and a run long
    enough for the reference's estimate to put its clump on the BadList (burst.c:3341-3354)"""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for i in range(240):
        s = acgt[rng.integers(0, 4, size=int(rng.integers(300, 900)))].copy()
        if i % 4 == 1:
            m = rng.random(len(s)) < 0.03
            s[m] = np.frombuffer(b"RYKMSWBDHVN", np.uint8)[rng.integers(0, 11, size=int(m.sum()))]
        if i == 6:
            s[100:109] = np.frombuffer(b"BDHVBDHVB", np.uint8)          # 3^9 words per window around it
        if i == 9:
            s[50:90] = ord("N")
        if i == 21:
            s[100:500] = np.frombuffer(b"RY", np.uint8)[rng.integers(0, 2, size=400)]     # estimate 3^11 per window x 400: BadList
        recs.append(s.tobytes().decode())
    fa = str(tmp_path / "amb.fa")
    with open(fa, "w") as f:
        for i, s in enumerate(recs):
            f.write(">r%d\n%s\n" % (i, s))
    for z in (1, 0):
        n, nbad, ndiff = _compare_built_with_loaded(None, 12, z, from_fasta=fa)
        assert 1 <= nbad <= 4 and n > 50000 and ndiff > 0, (n, nbad, ndiff)
    monkeypatch.setenv("BHIP_MASK_SLICE", "30000")
    _compare_built_with_loaded(None, 12, 1, from_fasta=fa)


@pytest.mark.parametrize("name", ["dna_q100_allpaths_fr", "dna_q100_best", "dna_q100_allpaths_y", "quick_q100_capitalist_fr", "quick_q292_best_fr", "dna_q292_forage_fr"])
def test_cli_with_device_built_accelerator(name, tmp_path):
    """burst_hip -ad: no .acx file anywhere -- golden outputs of the accelerated cases"""
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    cmd = [CLI, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"], "-ad"] + gl.cli_extra(c)

    def run(extra=()):
        r = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "built on the device" in r.stdout, r.stdout
        return sorted(open(out, "rb").read().splitlines())
    got = run()
    nd = run(["--no-dupe-hunt"]) if gl.order_sensitive(c) else None
    gl.compare(c, got, nd)


def _export(dev, K):
    lens, clumps, masks, bad = dev.acx_export(K)
    return lens, clumps, masks, bad


@pytest.mark.parametrize("db,K,z,n_ranks,slice_items", [("dna", 12, 1, 2, 0), ("dna", 12, 0, 3, 30000), ("quick", 15, 1, 2, 0), ("quick", 10, 1, 4, 0), ("quick", 12, 1, 5, 50000),
                                                        ("dna", 6, 1, 3, 0)])
def test_cooperative_build_equals_the_single_rank_build(db, K, z, n_ranks, slice_items, monkeypatch):
    """bhip_build_accelerator_shared: n_ranks handles of the replicated database (here: on one device, one thread each, as burst_hip --gpus N
    --devices 0,0,.. runs them) build the lists of their shares of the words and complete each other's tables (bhip_team_share) -- every
    handle must end up with the tables of the single-rank build, bit for bit: list lengths, clump ids in list order, lane sets, BadList.
    The single-rank tables come from the clump-sliced builder; the word-sliced one alone (the default; the cooperative builder with one
    rank) is compared on the way."""
    from burst_amd import host
    d = host.Db.read(os.path.join(gl.G, db + ".edx"))
    monkeypatch.setenv("BHIP_ACX_BUILD", "clumps")
    solo = d.open_device(0, z, build_K=K)
    want = _export(solo, K)
    solo.close()
    assert len(want[1]) > 1000
    if slice_items:
        monkeypatch.setenv("BHIP_MASK_SLICE", str(slice_items))      # (several slices inside a rank's share)
    monkeypatch.setenv("BHIP_ACX_BUILD", "words")
    w = d.open_device(0, z, build_K=K)
    got = _export(w, K)
    w.close()
    monkeypatch.delenv("BHIP_ACX_BUILD")
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    devs = d.open_devices_team([0] * n_ranks, z, build_K=K)
    for r, dev in enumerate(devs):
        got = _export(dev, K)
        for name, a, b in zip(("list lengths", "clump ids", "lane sets", "BadList"), want, got):
            bad = np.flatnonzero(a != b) if a.shape == b.shape else None
            assert np.array_equal(a, b), "rank %d's %s differ at %s of %d: want %s, got %s" % (r, name, None if bad is None else (bad[:5], bad[-5:], len(bad)), len(a),
                                                                                               None if bad is None else a[bad[:8]], None if bad is None else b[bad[:8]])
        dev.close()
    d.close()


def test_cooperative_build_falls_back_together(monkeypatch):
    """a rank that cannot do its share announces it in the first exchange and EVERY rank builds alone (nobody waits, same tables)"""
    from burst_amd import host
    d = host.Db.read(os.path.join(gl.G, "quick.edx"))
    monkeypatch.setenv("BHIP_ACX_BUILD", "clumps")
    solo = d.open_device(0, 1, build_K=12)
    want = _export(solo, 12)
    solo.close()
    monkeypatch.delenv("BHIP_ACX_BUILD")
    monkeypatch.setenv("BHIP_TEST_COOP_FAIL_RANK", "1")
    devs = d.open_devices_team([0, 0, 0], 1, build_K=12)
    for dev in devs:
        for a, b in zip(want, _export(dev, 12)):
            assert np.array_equal(a, b)
        dev.close()
    d.close()


@pytest.mark.parametrize("name,n", [("dna_q100_allpaths_fr", 2), ("quick_q100_capitalist_fr", 3), ("quick_q292_best_fr", 2)])
def test_cli_ranks_build_the_accelerator_together(name, n, tmp_path):
    """burst_hip --gpus N --devices 0,0,.. -ad: the ranks of a replicated database build ONE accelerator between them; golden outputs"""
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    cmd = [CLI, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"], "-ad", "--gpus", str(n), "--devices", ",".join(["0"] * n), "--gather", "host"] + gl.cli_extra(c)

    def run(extra=()):
        r = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, BHIP_DEBUG="1"))
        assert r.returncode == 0 and "built by the %d ranks together" % n in r.stdout and r.stdout.count("(cooperative)") == n, r.stdout
        return sorted(open(out, "rb").read().splitlines())
    got = run()
    nd = run(["--no-dupe-hunt"]) if gl.order_sensitive(c) else None
    gl.compare(c, got, nd)


def test_launcher_ranks_build_the_accelerator_together(tmp_path):
    """python -m burst_amd.run -ad with two processes (gloo; both on device 0): the exchange goes through the launcher's process group
    (host.dist_share), the .b6 is the one-process run's"""
    c = [x for x in gl.cases() if x["name"] == "quick_q100_capitalist_fr"][0]
    ref, q, fr, z, shear = gl.case_args(c)
    outs = []
    for world in (1, 2):
        out = str(tmp_path / ("o%d.b6" % world))
        args = ["-m", "burst_amd.run", "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"], "-ad"] + (["-fr"] if fr else [])
        cmd = [sys.executable] + args if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                          "--master-port", "29771"] + args
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd=gl.ROOT,
                           env=dict(os.environ, BURST_RUN_DEVICE="0", BHIP_DEBUG="1", PYTHONPATH=gl.ROOT))
        assert r.returncode == 0, r.stdout[-3000:]
        if world == 2:
            assert r.stdout.count("(cooperative)") == 2, r.stdout[-3000:]
        outs.append(sorted(open(out, "rb").read().splitlines()))
    assert outs[0] == outs[1] and len(outs[0]) > 100
    if not gl.order_sensitive(c):
        gl.compare(c, outs[1], None)


def test_device_builder_differential_against_the_reference(tmp_path):
    """tools/db_diff.py on the GPU box: `burst_hip -d QUICK ... -a` now builds the accelerator on the device; its .acx must be
    the compiled reference's, byte for byte, on random and awkward FASTA files (heavy IUPAC, N runs, -y, duplicates, ...)"""
    if not os.path.exists(os.path.join(gl.ROOT, "oracle", "_ref", "burst12")):
        pytest.skip("compiled reference not present")
    r = subprocess.run([sys.executable, os.path.join(gl.ROOT, "tools", "db_diff.py"), "6", str(tmp_path)], env=dict(os.environ, DB_DIFF_EXPECT_DEVICE="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("build", [True, False])
def test_record_lane_sets_are_the_coded_exact_masks(build):
    """the 4-byte records carry a lane-SET CODE: exact for one and two lanes, the smallest enclosing quad pattern beyond
    (burst_amd/csrc/bhip_lanecode.h).  Families of 1 .. 6 near-identical sequences inside a clump make every class of set occur;
    the exported masks must be exactly decode(encode(true lane mask)) -- built on the device and derived for a loaded .acx."""
    import dbutil
    import oraclelib as ol
    from burst_amd import capi, synth
    rng = np.random.default_rng(17)
    seqs = []
    while len(seqs) < 16 * 24:
        base = rng.integers(1, 5, size=260, dtype=np.uint8)
        seqs += synth.mutate_family(base, int(rng.integers(1, 7)), 0.02, rng)
    seqs = seqs[:16 * 24]
    order = rng.permutation(len(seqs))          # families scattered over the lanes of their clumps
    seqs = [seqs[i] for i in order]
    K = 10
    packed, clump_len, tot = dbutil.pack_clumps(seqs)
    lens, entries, offs = dbutil.build_acx(seqs, K)
    # true lane masks per (word, clump) in list order
    pairs = {}
    for i, s in enumerate(seqs):
        s = np.asarray(s, np.int64) - 1
        w = np.zeros(len(s) - K + 1, np.int64)
        for k in range(K):
            w = (w << 2) | s[k:len(s) - K + 1 + k]
        for x in np.unique(w):
            pairs[(int(x), i // 16)] = pairs.get((int(x), i // 16), 0) | (1 << (i % 16))
    words = np.repeat(np.arange(len(lens)), lens)
    true = np.array([pairs[(int(w), int(c))] for w, c in zip(words, entries)], np.uint32)

    def enc(m):
        pc = bin(m).count("1")
        if pc <= 2:
            return m
        qm = [q for q in range(4) if (m >> (4 * q)) & 15]
        if len(qm) == 1:
            sub = (m >> (4 * qm[0])) & 15
            return (sub if bin(sub).count("1") == 3 else 15) << (4 * qm[0])
        return sum(15 << (4 * q) for q in qm)
    want = np.array([enc(int(m)) for m in true], np.uint32)
    lut = ol.score_lut(1)
    if build:
        dev = capi.Device(packed, clump_len, tot, lut, K=K, build_acx=True)
    else:
        dev = capi.Device(packed, clump_len, tot, lut, acx_lens=lens, acx_lists=dbutil.pack_acx_lists(lens, entries, 1), acx_fmt=1, K=K)
    lens_d, clumps_d, masks_d, _ = dev.acx_export(K)
    dev.close()
    assert np.array_equal(lens_d, lens) and np.array_equal(clumps_d, entries)
    assert np.array_equal(masks_d.astype(np.uint32), want)
    # every class occurs: single lanes, pairs, quad patterns, unions of quads
    pcs = np.array([bin(int(m)).count("1") for m in true])
    assert (pcs == 1).any() and (pcs == 2).any() and (pcs >= 3).any() and (want != true).any()
