"""End to end on the MI355X: the burst_hip command line (C host + libburst_hip.so) against the reference's golden
.b6 for every case in tests/golden/cases.json -- all modes, with/without accelerator, -fr, -y, direct FASTA."""
import os
import subprocess

import numpy as np
import pytest

import goldenlib as gl

pytestmark = pytest.mark.gpu
CLI = os.path.join(gl.ROOT, "burst_amd", "burst_hip")
_acx = {}


def acx_for(db, z, tmpdir):
    key = (db, z)
    if key not in _acx:
        path = os.path.join(tmpdir, "%s_%d.acx" % (db, z))
        cmd = [CLI, "-r", os.path.join(gl.G, db + ".edx"), "--make-acx", path] + ([] if z else ["-y"])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        _acx[key] = path
    return _acx[key]


@pytest.mark.parametrize("c", gl.cases(), ids=lambda c: c["name"])
def test_cli_matches_reference(c, tmp_path_factory):
    tmp = str(tmp_path_factory.getbasetemp())
    ref, q, fr, z, shear = gl.case_args(c)
    out = os.path.join(tmp, c["name"] + ".out")
    cmd = [CLI, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"]] + gl.cli_extra(c)
    if c["accel"] and c["db"] != "fasta":
        cmd += ["-a", acx_for(c["db"], z, tmp)]

    def run(extra=(), mode=None):
        r = subprocess.run([mode if (mode and a == c["mode"] and cmd[i - 1] == "-m") else a for i, a in enumerate(cmd)] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        return sorted(open(out, "rb").read().splitlines())
    got = run()
    # (ANY: every placement within budget = what FORAGE prints without the duplicate hunt)
    nd = run(["--no-dupe-hunt"], "FORAGE" if c["mode"] == "ANY" else None) if gl.order_sensitive(c) else None
    gl.compare(c, got, nd)


def test_cli_small_batches_equal_one_batch(tmp_path):
    """the batch scheduler must not change results: 37 unique queries per device call vs one call"""
    c = [x for x in gl.cases() if x["name"] == "dna_q100_allpaths_fr"][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    acx = acx_for("dna", 1, str(tmp_path))
    subprocess.check_call([CLI, "-r", ref, "-a", acx, "-q", q, "-o", out, "-m", "ALLPATHS", "-i", "0.95", "-fr", "--batch", "37"], stdout=subprocess.DEVNULL)
    assert sorted(open(out, "rb").read().splitlines()) == gl.golden_lines(c)


def test_cli_error_codes(tmp_path):
    out = str(tmp_path / "o.b6")
    r = subprocess.run([CLI, "-r", os.path.join(gl.G, "dna.edx"), "-q", "/nonexistent.fa", "-o", out])
    assert r.returncode == 2            # cannot open queries (burst.c:639)
    r = subprocess.run([CLI, "-r", os.path.join(gl.G, "dna.edx"), "-q", os.path.join(gl.G, "q100.fa"), "-o", out, "-m", "NOPE"], stdout=subprocess.DEVNULL)
    assert r.returncode == 1            # usage (burst.c:4966)


def test_cli_k15_accelerator(tmp_path_factory):
    """DB15 (burst.c:96-99, the build the reference ships for large databases): 15-mers, a 4 GiB length table.  The
    accelerator our builder writes for the QUICK database has the sha256 of the one the reference (compiled with
    -DSCOUR_N=15) wrote, and alignments through it give the golden outputs (which do not depend on K)."""
    import hashlib
    import json
    tmp = str(tmp_path_factory.getbasetemp())
    acx = os.path.join(tmp, "quick_k15.acx")
    subprocess.check_call([CLI, "-r", os.path.join(gl.G, "quick.edx"), "--make-acx", acx, "-k", "15"], stdout=subprocess.DEVNULL)
    h = hashlib.sha256()
    with open(acx, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert h.hexdigest() == json.load(open(os.path.join(gl.G, "acx.sha256")))["quick_k15.acx"]
    try:
        for name in ("quick_q100_capitalist_fr", "quick_q292_best_fr", "quick_q100_forage"):
            c = [x for x in gl.cases() if x["name"] == name][0]
            ref, q, fr, z, shear = gl.case_args(c)
            out = os.path.join(tmp, name + ".k15.out")
            cmd = [CLI, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"], "-a", acx, "-k", "15"] + gl.cli_extra(c)

            def run(extra=()):
                r = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                assert r.returncode == 0, r.stdout
                assert "K=15" in r.stdout
                return sorted(open(out, "rb").read().splitlines())
            got = run()
            nd = run(["--no-dupe-hunt"]) if gl.order_sensitive(c) else None
            gl.compare(c, got, nd)
    finally:
        os.remove(acx)


@pytest.mark.parametrize("mode,ident", [("ALLPATHS", 0.95), ("FORAGE", 0.95), ("BEST", 0.97)])
def test_database_shards_on_one_device_equal_whole(mode, ident):
    """database sharding (burst_amd.run --shard db) without a second GPU: the clump ranges three ranks would hold are
    searched one after the other through the device path (bh_db_slice -> bhip_init -> bh_align), the per-query minimum is
    combined as the all_reduce would, and the surviving records must be byte-identical to the whole database's"""
    import ctypes as C
    import numpy as np
    from burst_amd import capi, host
    db = host.Db.read(os.path.join(gl.G, "dna.edx"))
    host._chk(host.lib().bh_acx_build(C.byref(db.c), 12, 1))
    qs = host.QuerySet(os.path.join(gl.G, "q100.fa"), ident, rc=True, accel=True, K=12)
    L = host.lib()

    def search(part):
        dev = part.open_device(0, 1)
        run = host.BhRun()
        host._chk(L.bh_align(dev._h, C.byref(qs.c), 0, qs.n_uniq, host.MODES[mode], 1 << 18, C.byref(run)))
        n = int(run.nHits)
        out = np.ctypeslib.as_array(C.cast(run.hits, C.POINTER(C.c_uint8)), shape=(n * 20,)).view(capi.HIT_DTYPE).copy() if n else np.zeros(0, capi.HIT_DTYPE)
        L.bh_run_free(C.byref(run))
        dev.close()
        return out
    whole = search(db)
    assert len(whole) > 100
    cl = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    six = host._view(qs.c.six, qs.n_entries, np.uint32)
    parts = []
    for rank in range(3):
        c0, c1 = host.clump_shard(db, 3, rank)
        h = search(db.slice(c0, c1))
        h["refIx"] += np.uint32(16 * c0)
        parts.append(h)
    if mode != "FORAGE":      # what the all_reduce(MIN) of the ranks' per-query minima does, and the filter behind it
        gmin = np.full(qs.n_uniq, 255, np.uint8)
        for h in parts:
            np.minimum.at(gmin, six[h["q"]], h["ed"])
        parts = [h[h["ed"] == gmin[six[h["q"]]]] for h in parts]
    got = np.concatenate(parts)
    got = got[np.lexsort((got["refIx"], got["q"]))]
    if mode == "BEST":
        # -m BEST keeps ONE record per entry on the device (BHIP_HITS_BEST: higher score, then lower RefIxSrt, burst.c:4847-4891), every shard
        # its own; of the shards' survivors the same rule must leave the whole database's choice
        order = host._view(db.c.refIxSrt, db.c.totR, np.uint32)
        pick = np.lexsort((order[got["refIx"]], -got["score"].astype(np.float64), got["q"]))
        got = got[pick]
        got = got[np.concatenate(([True], got["q"][1:] != got["q"][:-1]))]
        assert len(whole) == len(np.unique(whole["q"]))
    assert got.tobytes() == whole[np.lexsort((whole["refIx"], whole["q"]))].tobytes()


def test_cli_differential_on_awkward_inputs(tmp_path):
    """tools/cli_diff.py: burst_hip next to the compiled reference (oracle/_ref/burst12, built by __graft_entry__.build()
    where /root/reference exists; it travels with the snapshot) on query files with lower case, CRLF, duplicates, very short
    reads, N runs, IUPAC codes, symbols outside the alphabet, long and mixed-length reads -- every mode, same lines, same exit
    codes (including the reference's own stop on a query that starts with a symbol outside the alphabet)."""
    import sys
    if not os.path.exists(os.path.join(gl.ROOT, "oracle", "_ref", "burst12")):
        pytest.skip("compiled reference not present")
    r = subprocess.run([sys.executable, os.path.join(gl.ROOT, "tools", "cli_diff.py"), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "ALL OK" in r.stdout


def test_command_line_with_reads_beyond_1024_symbols(tmp_path):
    """burst_hip next to the compiled reference (-t 1: its deterministic configuration) on reads of 1 100 .. 4 000 symbols mixed with
    100-symbol reads, against unsheared 5 000-symbol references searched directly and through a database with an accelerator"""
    import numpy as np
    ref_exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12")
    cli = os.path.join(gl.ROOT, "burst_amd", "burst_hip")
    if not os.path.exists(ref_exe):
        pytest.skip("compiled reference not present")
    rng = np.random.default_rng(77)
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
    for f in range(4):
        base = "".join(A[rng.integers(0, 4, size=5000)])
        refs += [("fam%d_v%d" % (f, v), mutate(base, 40 * v)) for v in range(5)]
    refs_fa = str(tmp_path / "refs.fa")
    open(refs_fa, "w").write("".join(">%s\n%s\n" % r for r in refs))
    reads = []
    for i in range(36):
        h, s = refs[int(rng.integers(len(refs)))]
        n = 100 if i % 3 == 2 else int(rng.integers(1100, 4001))
        st = int(rng.integers(0, len(s) - n))
        reads.append(("r%d_%s" % (i, h), mutate(s[st:st + n], int(rng.integers(0, max(2, n // 60))))))
    reads.append(("ambig_long", reads[0][1][:700] + "N" + reads[0][1][701:1500] + "R" + reads[0][1][1501:]))
    q_fa = str(tmp_path / "q.fa")
    open(q_fa, "w").write("".join(">%s\n%s\n" % r for r in reads))
    edx, acx = str(tmp_path / "db.edx"), str(tmp_path / "db.acx")
    subprocess.check_call([ref_exe, "-r", refs_fa, "-d", "QUICK", "4200", "-o", edx, "-a", acx, "-t", "1", "--noprogress"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n_lines = 0
    for mode, ident, extra in (("BEST", "0.97", ["-r", refs_fa]), ("ALLPATHS", "0.96", ["-r", refs_fa, "-fr"]), ("CAPITALIST", "0.97", ["-r", refs_fa, "-fr"]),
                               ("FORAGE", "0.95", ["-r", refs_fa]), ("BEST", "0.97", ["-r", edx, "-a", acx]), ("ALLPATHS", "0.96", ["-r", edx, "-a", acx, "-fr"])):
        outs = []
        for exe, tail in ((ref_exe, ["-t", "1", "--noprogress"]), (cli, [])):
            o = str(tmp_path / ("out_%s.b6" % ("ref" if exe == ref_exe else "hip")))
            if os.path.exists(o):
                os.remove(o)
            r = subprocess.run([exe, "-q", q_fa, "-o", o, "-m", mode, "-i", ident] + extra + tail, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, (exe, mode, r.stdout[-1500:])
            outs.append(sorted(open(o, "rb").read().splitlines()))
        assert outs[0] == outs[1], (mode, extra[2:], len(outs[0]), len(outs[1]), sorted(set(outs[0]) ^ set(outs[1]))[:4])
        assert len(outs[0]) >= 30
        n_lines += len(outs[0])
    assert n_lines > 300


def test_python_launcher_single_process(tmp_path):
    """python -m burst_amd.run (the multi-GPU front end) with one process is equivalent to burst_hip"""
    import sys
    c = [x for x in gl.cases() if x["name"] == "dna_q100_allpaths_fr"][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    env = dict(os.environ, PYTHONPATH=gl.ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "burst_amd.run", "-r", ref, "-a", acx_for("dna", 1, str(tmp_path)), "-q", q, "-o", out, "-m", "ALLPATHS", "-i", "0.95", "-fr"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=gl.ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert sorted(open(out, "rb").read().splitlines()) == gl.golden_lines(c)


@pytest.mark.parametrize("name,world", [("dna_q100_capitalist_fr", 2), ("dna_q292_forage_fr", 3)])
def test_python_launcher_processes_share_the_records_in_shared_memory(name, world, tmp_path):
    """python -m burst_amd.run under torch.distributed.run with several processes (BURST_RUN_DEVICE=0: all on the one device of the
    test box): every rank aligns its share of the unique queries, its records lie in its shared-memory segment, rank 0 writes the
    report straight from the segments (bh_node.c + bh_report_view; CAPITALIST's vote runs over all of them): the golden lines"""
    import sys
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    env = dict(os.environ, PYTHONPATH=gl.ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), BURST_RUN_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29620 + world),
           "-m", "burst_amd.run", "-r", ref, "-a", acx_for(c["db"], z, str(tmp_path)), "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"]] + (["-fr"] if fr else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=gl.ROOT, timeout=600)
    assert r.returncode == 0 and "from %d rank(s)" % world in r.stdout, r.stdout[-3000:]
    gl.compare(c, sorted(open(out, "rb").read().splitlines()), None)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("burst_hip.run")]


@pytest.mark.parametrize("c", gl.cases(), ids=lambda c: c["name"])
def test_reference_host_with_device_binding(c, tmp_path_factory):
    """INTEGRATION.md made executable: oracle/_ref/burst12_hip is the reference's own burst.c with oracle/burst_hip_binding.inc
    spliced into do_alignments (built by oracle/make_ref_hip.py in the build container) -- its parser, query pipeline, database
    and FASTA readers, consolidation and .b6 writer, with bhip_init / bhip_align_batch in place of the two OpenMP loops.  Every
    golden case goes through it -- the five run modes (ANY printed by the binding itself, as the reference's loops do), with and
    without accelerator, FASTA references, taxonomy -- and its output must be the golden output of the unmodified reference
    (the cases whose reference output depends on its threads' hit order: under the relaxed contract of goldenlib.compare)."""
    exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12_hip")
    if not os.path.exists(exe):
        pytest.skip("patched reference not built (oracle/make_ref_hip.py needs /root/reference)")
    tmp = str(tmp_path_factory.getbasetemp())
    ref, q, fr, z, shear = gl.case_args(c)
    out = os.path.join(tmp, c["name"] + ".refhip.out")
    cmd = [exe, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"], "-t", str(c["threads"]), "--noprogress"] + gl.cli_extra(c)
    acx = ["-a", acx_for(c["db"], z, tmp)] if c["accel"] and c["db"] != "fasta" else []
    r = subprocess.run(cmd + acx, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "records from the device path" in r.stdout, r.stdout[-2000:]
    nd = None
    if gl.order_sensitive(c):      # every placement within budget, from burst_hip itself (ANY: what FORAGE prints without the duplicate hunt)
        nd_out = os.path.join(tmp, c["name"] + ".refhip.nd")
        subprocess.check_call([CLI, "-r", ref, "-q", q, "-o", nd_out, "-m", "FORAGE" if c["mode"] == "ANY" else c["mode"], "-i", c["id"], "--no-dupe-hunt"] + gl.cli_extra(c) + acx, stdout=subprocess.DEVNULL)
        nd = sorted(open(nd_out, "rb").read().splitlines())
    gl.compare(c, sorted(open(out, "rb").read().splitlines()), nd)


@pytest.mark.parametrize("name,flags,expect", [("dna_q100_best_fr", ["--gpus", "1", "--gather", "rccl"], "RCCL gather: 1 rank(s)"),
                                               ("dna_q100_allpaths_fr", ["--gpus", "1", "--batch", "29", "--gather", "rccl"], "RCCL gather: 1 rank(s)"),
                                               ("dna_q100_allpaths_fr", ["--gpus", "3", "--devices", "0,0,0", "--gather", "host"], "host gather: 3 rank(s)"),
                                               ("dna_q100_capitalist_noacx_t1_fr", ["--gpus", "2", "--devices", "0,0", "--gather", "host"], "host gather: 2 rank(s)"),
                                               ("dna_q100_allpaths_fr", ["--gpus", "3", "--devices", "0,0,0", "--gather", "host", "--shard", "db"], "host gather: 3 rank(s), database-sharded"),
                                               ("dna_q100_best_fr", ["--gpus", "4", "--devices", "0,0,0,0", "--gather", "host", "--shard", "db", "-ad"], "host gather: 4 rank(s), database-sharded"),
                                               ("dna_q100_capitalist_fr", ["--gpus", "2", "--devices", "0,0", "--gather", "host", "--shard", "db", "--batch", "41"], "host gather: 2 rank(s), database-sharded"),
                                               ("dna_q292_forage_fr", ["--gpus", "3", "--devices", "0,0,0", "--gather", "host", "--shard", "db", "-ad"], "host gather: 3 rank(s), database-sharded"),
                                               ("quick_q100_capitalist_noacx_t1", ["--gpus", "2", "--devices", "0,0", "--gather", "host", "--shard", "db"], "host gather: 2 rank(s), database-sharded"),
                                               ("dna_q100_allpaths_fr", ["--gpus", "4", "--devices", "0,0,0,0", "--gather", "host", "--shards", "2"], "host gather: 4 rank(s), database-sharded"),
                                               ("dna_q100_capitalist_fr", ["--gpus", "6", "--devices", "0,0,0,0,0,0", "--gather", "host", "--shards", "3", "-ad", "--batch", "53"], "host gather: 6 rank(s), database-sharded"),
                                               ("dna_q100_allpaths_fr", ["--gpus", "1", "--shard", "db", "--gather", "rccl"], "RCCL gather: 1 rank(s)"),
                                               # more shards than devices: the shards take turns on the one device (bh_search_serial_shards)
                                               ("dna_q100_allpaths_fr", ["--gpus", "1", "--shards", "3"], "serial shards: 3 shard(s) taking turns"),
                                               ("dna_q100_capitalist_fr", ["--gpus", "1", "--shards", "2", "-ad", "--batch", "37"], "serial shards: 2 shard(s) taking turns"),
                                               ("dna_q292_forage_fr", ["--gpus", "1", "--shards", "4", "-ad"], "serial shards: 4 shard(s) taking turns"),
                                               ("dna_q100_best_tax_fr", ["--gpus", "1", "--shards", "2"], "serial shards: 2 shard(s) taking turns")])
def test_cli_multi_gpu_paths(name, flags, expect, tmp_path):
    """burst_hip --gpus N: one host thread + one device handle per rank, the unique queries sharded, the records gathered to
    rank 0 (ncclAllGather of the counts + grouped ncclSend / ncclRecv in libburst_hip; `--gather host` when the ranks share a
    device, as they must on a one-GPU box).  --gpus 1 goes through the RCCL code with one rank.  Same .b6 as one device.
    --gpus 1 --shards S: a database of S shards on one device, the shards taking turns (bh_search_serial_shards).
    --shard db (bh_search_multi): every rank holds a range of clumps -- the .acx lists restricted to it (bh_db_slice), or with -ad an
    accelerator the rank's device builds for its slice -- aligns all queries, the per-query minimum is combined over the ranks and
    the gathered records are put in (query, reference) order: the golden lines of the whole database.  --shards S with more ranks:
    replica groups of S shards, the queries cut over the groups."""
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    cmd = [CLI, "-r", ref, "-q", q, "-o", out, "-m", c["mode"], "-i", c["id"]] + gl.cli_extra(c) + flags
    if c["accel"] and "-ad" not in flags:
        cmd += ["-a", acx_for(c["db"], z, str(tmp_path))]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and expect in r.stdout, r.stdout[-2000:]
    got = open(out, "rb").read()
    if "--shard" in flags or "--shards" in flags:
        # the record set and its order are those of one device holding the whole database: the .b6 must be the single-device run's,
        # byte for byte in the same order (also in the modes whose golden depends on the reference's thread timing)
        single = [x for x in cmd if x not in ("--shard", "db", "--gather", "host", "rccl")]
        for opt in ("--gpus", "--devices", "--batch", "--shards"):
            if opt in single:
                k = single.index(opt)
                del single[k:k + 2]
        r1 = subprocess.run(single, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r1.returncode == 0, r1.stdout[-2000:]
        assert open(out, "rb").read() == got
        gl.compare(c, sorted(got.splitlines()), None)
    else:
        assert sorted(got.splitlines()) == gl.golden_lines(c)


def test_cli_reads_fastq_gz(tmp_path):
    """a gzip-compressed FASTQ file of the golden reads gives the golden lines (the reference, which reads neither, was run on the
    same reads as two-line FASTA)"""
    import gzip
    c = [x for x in gl.cases() if x["name"] == "dna_q100_allpaths_fr"][0]
    ref, q, fr, z, shear = gl.case_args(c)
    recs = [r for r in open(q).read().split("\n") if r != ""]
    fq = str(tmp_path / "q.fastq.gz")
    with gzip.open(fq, "wt") as g:
        for h, s in zip(recs[0::2], recs[1::2]):
            g.write("@%s\n%s\n+\n%s\n" % (h[1:], s, "F" * len(s)))
    out = str(tmp_path / "o.b6")
    subprocess.check_call([CLI, "-r", ref, "-a", acx_for("dna", 1, str(tmp_path)), "-q", fq, "-o", out, "-m", "ALLPATHS", "-i", "0.95", "-fr"], stdout=subprocess.DEVNULL)
    assert sorted(open(out, "rb").read().splitlines()) == gl.golden_lines(c)


@pytest.mark.gpu
def test_device_query_sort_equals_host_sort(tmp_path, monkeypatch, capfd):
    """bhip_sort_queries (LSD radix sort of the query records on the device + duplicate marking) against the host's sort of
    the same file: 300 k reads of mixed lengths with exact duplicates, records that are prefixes of others, IUPAC codes,
    symbols outside the alphabet (code 0) -- every table the loader builds (order of the headers, offsets of the unique
    queries, lengths, budgets, symbol codes, strands) must be identical, with and without reverse complements."""
    import ctypes as C
    from burst_amd import host
    rng = np.random.default_rng(77)
    n = 300_000
    alpha = np.frombuffer(b"ACGT", np.uint8)
    base = [alpha[rng.integers(0, 4, int(L))] for L in rng.integers(30, 260, 4000)]
    recs = []
    for i in range(n):
        b = base[int(rng.integers(0, len(base)))]
        k = rng.random()
        if k < 0.35:
            s = b.copy()                                         # exact duplicate of a base
        elif k < 0.6:
            s = b[:int(rng.integers(1, len(b) + 1))].copy()      # a prefix of it
        else:
            s = b.copy()
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(0, len(s)))] = alpha[int(rng.integers(0, 4))]
        if k > 0.97:
            s[int(rng.integers(0, len(s)))] = ord("NRYK@-"[int(rng.integers(0, 6))])       # IUPAC, and symbols of code 0
        recs.append(b">r%d\n" % i + s.tobytes() + b"\n")
    fa = tmp_path / "mixed.fa"
    fa.write_bytes(b"".join(recs))

    def load(host_sort, rc):
        if host_sort:
            monkeypatch.setenv("BURST_HOST_SORT", "1")
        else:
            monkeypatch.delenv("BURST_HOST_SORT", raising=False)
        qs = host.QuerySet(str(fa), 0.97, rc=rc, accel=True, K=12)
        c = qs.c
        heads = np.ctypeslib.as_array(C.cast(c.heads, C.POINTER(C.c_uint64)), (qs.n_reads,)).copy() - int(c.dump or 0)
        qoff = host._view(c.qoff, qs.n_entries + 1, np.uint64).copy()
        out = dict(n=(qs.n_reads, qs.n_uniq, qs.n_entries, int(c.maxLen), int(c.minLen), int(c.maxED), int(c.nClear), int(c.nAmbig), int(c.nBad)),
                   heads=heads, offset=host._view(c.offset, qs.n_uniq + 1, np.uint64).copy(), qoff=qoff,
                   codes=host._view(c.codes, int(qoff[-1]), np.uint8).copy(), len=host._view(c.len, qs.n_uniq, np.uint32).copy(),
                   emac=host._view(c.emac, qs.n_entries, np.uint16).copy(), rc=host._view(c.rc, qs.n_entries, np.uint8).copy(),
                   flags=host._view(c.flags, qs.n_entries, np.uint8).copy())
        qs.close()
        return out
    monkeypatch.setenv("BURST_HOST_DEBUG", "1")
    for rc in (False, True):
        a = load(True, rc)
        capfd.readouterr()
        b = load(False, rc)
        err = capfd.readouterr().err
        assert "sort + duplicates (device)" in err and "not available" not in err, err       # the device path really ran
        assert a["n"] == b["n"] and a["n"][1] < a["n"][0]          # duplicates were folded
        for k in a:
            if k != "n":
                assert np.array_equal(a[k], b[k]), (rc, k)


def test_bench_multi_rank_path_with_one_process(tmp_path):
    """bench.py's N > 1 code paths on the one GPU of the test box, on a small database; each JSON line must carry the same record
    count as the plain single-process run of the same job:
    * --gather rccl with one process (BURST_BENCH_DIST1=1 under torch.distributed.run): the library's RCCL communicator made from a
      broadcast id (bhip_comm_unique_id / bhip_comm_create_rank) and bh_search_multi -- align + bhip_comm_gather_hits in the timed region;
    * the default hand-over with the driver's own command line for N = 2 (two processes, BURST_BENCH_DEVICE=0 puts both ranks on the
      one device), strong and weak scaling: every rank's records in its shared-memory segment, rank 0 reads them there (bh_node.c)"""
    import json
    import sys
    bench = os.path.join(gl.ROOT, "bench.py")
    common = ["--db-scale", "1", "--n-base", "20000", "--reads", "100000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end", "--no-continuity", "--workdir", str(tmp_path / "w")]
    r1 = subprocess.run([sys.executable, bench] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a = json.loads(r1.stdout.strip().splitlines()[-1])
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--master-addr", "127.0.0.1"]
    r2 = subprocess.run(launch + ["--nproc-per-node", "1", "--master-port", "29611", bench, "--gpus", "1", "--gather", "rccl"] + common,
                        env=dict(os.environ, BURST_BENCH_DIST1="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = json.loads([ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert a["work"]["records"] == b["work"]["records"] > 250000
    assert "RCCL gather" in b["config"]["parallelism"] and b["n_gpus"] == 1
    # the default line for N > 1 is the FIXED job (strong scaling), with the weak-scaling figure and configs[3]'s 10 M-read job as extra keys
    r3 = subprocess.run(launch + ["--nproc-per-node", "2", "--master-port", "29612", bench, "--gpus", "2"] + common,
                        env=dict(os.environ, BURST_BENCH_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr[-3000:]
    c = json.loads([ln for ln in r3.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert c["work"]["records"] == a["work"]["records"]
    assert "shared-memory" in c["config"]["parallelism"] and c["n_gpus"] == 2 and c["scaling"] == "strong"
    # rank 0 read every rank's records where they lie (no copy): two runs, together the single-process run's records
    assert c["handover"]["kind"].startswith("view") and len(c["handover"]["records_per_run"]) == 2 and sum(c["handover"]["records_per_run"]) == a["work"]["records"]
    assert min(c["handover"]["records_per_run"]) > 100000 and c["handover"]["distinct_entries"] > 100000
    assert abs(c["weak_scaling"]["reads"] - 2 * 300000) < 3000          # (a pool batch is a range of UNIQUE queries: about --reads reads)
    assert 1.9 * a["work"]["records"] < c["weak_scaling"]["records"] < 2.1 * a["work"]["records"]
    assert c["configs3_job"]["reads"] > 0 and c["configs3_job"]["records"] > 0 and c["configs3_job"]["value"] > 0
    assert "rccl" not in c          # (two ranks on one device: RCCL refuses; on a multi-GPU node the key is there, see r5)
    # --scaling weak: every rank its own three batches, twice the single-process job's reads and (about) records
    r4 = subprocess.run(launch + ["--nproc-per-node", "2", "--master-port", "29613", bench, "--gpus", "2", "--scaling", "weak"] + common,
                        env=dict(os.environ, BURST_BENCH_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r4.returncode == 0, r4.stderr[-3000:]
    d = json.loads([ln for ln in r4.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "weak" and d["n_gpus"] == 2 and len(d["handover"]["records_per_run"]) == 2
    assert 1.9 * a["work"]["records"] < d["work"]["records"] < 2.1 * a["work"]["records"] and min(d["handover"]["records_per_run"]) > 0.9 * a["work"]["records"]
    assert d["strong_scaling"]["records"] == a["work"]["records"]
    # the RCCL keys of the N > 1 line: the launcher's code path with one process (the only way to have a communicator on a one-GPU box):
    # the shared-memory hand-over is timed as `value`, then the same job with bhip_comm_gather_hits inside the timed region
    r5 = subprocess.run(launch + ["--nproc-per-node", "1", "--master-port", "29614", bench, "--gpus", "1"] + common,
                        env=dict(os.environ, BURST_BENCH_DIST1="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r5.returncode == 0, r5.stderr[-3000:]
    e = json.loads([ln for ln in r5.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert e["work"]["records"] == a["work"]["records"] and "shared-memory" in e["config"]["parallelism"]
    assert e["rccl"]["rccl_ranks"] == 1 and e["rccl"]["records"] == a["work"]["records"] and e["rccl"]["value"] > 0 and e["rccl"]["rccl_gather_ms"] >= 0
    # at the top level and in `config` too (what the driver's parsed view keeps), and gathered from the device-resident records
    assert e["rccl_ranks"] == 1 == e["config"]["rccl_ranks"] and e["rccl_gather_ms"] == e["rccl"]["rccl_gather_ms"] == e["config"]["rccl_gather_ms"]
    assert e["config"]["rccl_records_from"].startswith("device") and e["rccl"]["gather_path"] == 1      # (BhMultiRank.gatherPath: the staged gather really ran)
    # ... and which exchange built the accelerator: the bench's ranks each build their own, no collective (the line says so)
    assert e["config"]["accelerator_build"]["exchange"] is None and e["config"]["accelerator_build"]["ranks_in_exchange"] == 0
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("burst_hip.bench")]


@pytest.mark.parametrize("alphabet,mode", [("ACDEFGHIKLMN", "BEST"), ("ACDEFGHIKLMN", "ALLPATHS"), ("ACGTacgtN", "CAPITALIST"), ("01", "FORAGE")])
def test_cli_xalphabet_matches_the_reference(tmp_path, alphabet, mode):
    """burst_hip -x (any alphabet of up to 15 symbols, compared for equality: aded_xalpha / reScoreM_xalpha, burst.c:696-697, 894, 1099)
    on the device against the compiled reference with -x on the same sequences over bytes its query sort survives (tests/xalpha_util.py);
    more than 15 symbols, -fr and databases are refused as upstream cannot do them either"""
    ref_exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12")
    if not os.path.exists(ref_exe):
        pytest.skip("compiled reference not present")
    import xalpha_util
    rf, qf, rf_low, qf_low = xalpha_util.write_inputs(tmp_path, alphabet, len(alphabet) * 11 + len(mode), n_refs=60, n_queries=400)
    out_ref, out = str(tmp_path / "ref.b6"), str(tmp_path / "hip.b6")
    r = subprocess.run([ref_exe, "-x", "-r", rf_low, "-q", qf_low, "-o", out_ref, "-m", mode, "-i", "0.93", "-t", "1", "--noprogress"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-800:]
    r = subprocess.run([CLI, "-x", "-r", rf, "-q", qf, "-o", out, "-m", mode, "-i", "0.93"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-800:]
    a, b = sorted(open(out_ref, "rb").read().splitlines()), sorted(open(out, "rb").read().splitlines())
    assert len(a) > 250 and a == b
    # ... and the reference's own host with the binding spliced in (oracle/burst_hip_binding.inc maps the run's symbols onto the device's codes)
    hip_exe = os.path.join(gl.ROOT, "oracle", "_ref", "burst12_hip")
    if os.path.exists(hip_exe):
        out_b = str(tmp_path / "refhip.b6")
        r = subprocess.run([hip_exe, "-x", "-r", rf_low, "-q", qf_low, "-o", out_b, "-m", mode, "-i", "0.93", "-t", "1", "--noprogress"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "records from the device path" in r.stdout, r.stdout[-800:]
        assert sorted(open(out_b, "rb").read().splitlines()) == a
    # what upstream cannot do is refused, with its own message where it has one
    r = subprocess.run([CLI, "-x", "-r", os.path.join(gl.G, "dna.edx"), "-q", qf, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "Xalpha" in r.stdout
    r = subprocess.run([CLI, "-x", "-fr", "-r", rf, "-q", qf, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1
    with open(str(tmp_path / "wide.fa"), "wb") as f:
        f.write(b">w\nABCDEFGHIJKLMNOPQRSTUVWXYZ\n")
    r = subprocess.run([CLI, "-x", "-r", str(tmp_path / "wide.fa"), "-q", qf, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "distinct symbols" in r.stdout + (r.stderr or "")
