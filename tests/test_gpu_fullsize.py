"""BASELINE configs[1] at full size (1 M synthetic 100-bp reads vs the 99 k-reference GG97-like database of bench.py):
the oracle cannot finish this in reasonable time, so parity is checked through properties that do not depend on size --
determinism, invariance under every tuning option (the one-stage sweep, the LDS re-scorer and the clump-level prefilter
are independent implementations of the same result, and the lower-bound pruning must never drop a minimum), full sensitivity (every read carries <= 3 edits = its budget, so
every entry must be found with ed <= its number of edits), and the arithmetic identities of a record."""
import os
import re
import sys

import numpy as np
import pytest

CW_DEFAULT = 2          # library default of option prefilter_cw (see tests/test_gpu_kernels.py)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_million_reads_properties(tmp_path_factory):
    sys.path.insert(0, ROOT)
    import bench
    from burst_amd import host

    class A:
        pass
    a = A()
    a.read_len, a.n_base, a.n_variants, a.ref_len, a.variant_rate, a.id, a.K = 100, 3300, 30, 1400, 0.05, 0.97, 12
    work = os.environ.get("BURST_BENCH_DIR", "/tmp/burst_amd_bench")
    refs, edx, acx, done = bench.build_db(work, a)
    bench.ensure_acx(edx, acx, 12)
    reads = os.path.join(work, "fullsize_reads.fa")
    n_reads = 1000000
    if not os.path.exists(reads):
        host.synth_reads(refs, reads, n_reads, 100, [0, 1, 2, 3], rc=False, iupac=0.0, seed=4242)
    db = host.Db.read(edx, acx, K=12)
    qs = host.QuerySet(reads, 0.97, rc=False, accel=True, K=12)
    assert qs.n_reads == n_reads
    dev = db.open_device(0)
    q = qs.batch()
    dev.stage(q)
    base, _ = dev.align_staged(False)
    base = base.copy()
    # determinism
    again, _ = dev.align_staged(False)
    assert again.tobytes() == base.tobytes()
    # invariance under the tuning options (independent code paths)
    for opts in ({"lanes": 3}, {"prefilter_table": 10, "rescore_reg": 0}, {"lane_masks": 0}, {"two_stage": 0}, {"prefilter_stride": 6}, {"prune": 0},
                 {"prefilter_algo": 1}, {"prefilter_algo": 0, "prune": 1}, {"prefilter_cw": 1}, {"prefilter_cw": 2}, {"prefilter_cw": 0}):
        for k, v in opts.items():
            dev.set_option(k, v)
        if "lanes" in opts:
            dev.stage(q)                      # the lane cut is made at staging time
        got, _ = dev.align_staged(False)
        assert got.tobytes() == base.tobytes(), opts
        for k in opts:
            dev.set_option(k, {"lanes": 1, "prefilter_table": 0, "rescore_reg": 1, "lane_masks": 1, "two_stage": 1, "prefilter_stride": 0, "prune": 1, "prefilter_algo": -1, "prefilter_cw": CW_DEFAULT}[k])
        if "lanes" in opts:
            dev.stage(q)
    # sensitivity: a read carries at most 3 edits, so every entry whose budget is 3 must be found; the only entries that may
    # stay without a hit are the 97-symbol ones (three deletions) whose budget is 2 -- the exhaustive route confirms that
    # those have no alignment within budget (tools/missing_check.py)
    found = np.zeros(q.n, bool)
    found[base["q"]] = True
    assert (q.emac[~found] < 3).all() and (~found).sum() < 0.005 * q.n
    assert (base["ed"] <= q.emac[base["q"]]).all()
    best = np.full(q.n, 255, np.int64)
    np.minimum.at(best, base["q"], base["ed"])
    heads = []
    with open(reads) as f:
        for line in f:
            if line[0] == ">":
                heads.append(int(re.search(r"_e(\d+)", line).group(1)))
    assert len(heads) == n_reads
    assert best[found].mean() <= np.mean(heads) + 1e-9       # never worse than the edits that made the reads
    # record identities: 1 - ed / (len + gapQ) in f32, positions inside the clump, lanes below totR
    ln = np.diff(q.off.astype(np.int64))[base["q"]].astype(np.float32)
    want = (np.float32(1.0) - base["ed"].astype(np.float32) / (ln + base["gapQ"].astype(np.float32))).astype(np.float32)
    assert want.tobytes() == base["score"].astype(np.float32).tobytes()
    clump_len = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    assert (base["refIx"] < db.c.totR).all()
    assert (base["finalPos"] >= 1).all() and (base["finalPos"] <= clump_len[base["refIx"] >> 4]).all()
    order = np.lexsort((base["refIx"], base["q"]))
    assert (order == np.arange(len(base))).all()             # sorted by (query, refIx)
    dev.close()


def _bench_db(read_len, thres, K=12, n_base=3300, n_variants=30, want_acx=True):
    sys.path.insert(0, ROOT)
    import bench

    class A:
        pass
    a = A()
    a.read_len, a.n_base, a.n_variants, a.ref_len, a.variant_rate, a.id, a.K = read_len, n_base, n_variants, 1400, 0.05, thres, K
    work = os.environ.get("BURST_BENCH_DIR", "/tmp/burst_amd_bench")
    refs, edx, acx, done = bench.build_db(work, a)
    if want_acx:
        bench.ensure_acx(edx, acx, K)
    return work, refs, edx, acx


def _scale_diff(n_reads, env):
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_diff.sh"), str(n_reads)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500,
                       env=dict(os.environ, **env))
    return [ln for ln in r.stdout.splitlines() if ln.startswith(("BEST", "ALLPATHS", "CAPITALIST", "FORAGE"))], r.stdout


def _scale_diffs_side_by_side(jobs):
    """several tools/scale_diff.sh runs at once (jobs: (n_reads, env) -- a one-thread reference leaves 15 of the job's 16 cores idle; SD_TAG
    keeps their scratch files apart): the list of (lines, output) in the order given"""
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(jobs)) as pool:
        return list(pool.map(lambda j: _scale_diff(j[1][0], dict(j[1][1], SD_TAG="_%d" % j[0])), enumerate(jobs)))


def _check_diff_lines(lines, n_reads, frac=0.005):
    """BEST: identical.  The other modes print, where the reference's own thread timing decides (DUPE_HUNT between overlapping
    shears, burst.c:4563-4570; equally voted references in CAPITALIST, 4763-4776), one of several placements: there the
    contract is: same number of lines, same queries, at most 0.5 % of the lines differ (4 % for long reads on both strands in FORAGE,
    where a third of the reads lie across a shear boundary: the bound of tests/goldenlib.py) and every differing reference line is
    EXPLAINED -- one of the placements burst_hip computes for that query (--no-dupe-hunt prints them all; for CAPITALIST: one
    of the query's minimum placements)."""
    for ln in lines:
        if ln.startswith("BEST") or "IDENTICAL" in ln:
            assert "IDENTICAL" in ln, ln
            continue
        m = re.search(r"(\d+) of (\d+) lines differ", ln)
        n_diff, n_lines = int(m.group(1)), int(m.group(2))
        assert n_diff <= max(2, int(n_lines * frac)), ln
        assert "queries reported by only one program: 0;" in ln, ln
        a_, b_ = ln.split("line counts")[1].split("[")[0].split("/")
        assert int(a_) == int(b_), ln
        assert "not a placement burst_hip computed: 0;" in ln, ln


def test_reference_binary_parity_at_bench_size():
    """tools/scale_diff.sh: the compiled reference (oracle/_ref/burst12, all host threads, with the accelerator) and the
    burst_hip command line on 200 000 reads of the configs[1] workload, in all four modes the reference consolidates."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst12")):
        pytest.skip("compiled reference not present")
    work, refs, edx, acx = _bench_db(100, 0.97)
    from burst_amd import host
    reads = os.path.join(work, "reads_1000000_l100_e0-1-2-3_u0.0_f0_r0.fa")
    if not os.path.exists(reads):
        host.synth_reads(refs, reads, 1000000, 100, [0, 1, 2, 3], rc=False, iupac=0.0, seed=42)
    env = dict(BURST_BENCH_DIR=work, SD_EDX=edx, SD_READS=reads)
    lines, out = _scale_diff(200000, dict(env, SD_MODES="BEST ALLPATHS", SD_IDS="0.97 0.98"))
    assert len(lines) == 4, out[-3000:]
    _check_diff_lines(lines, 200000)
    lines, out = _scale_diff(200000, dict(env, SD_MODES="CAPITALIST FORAGE", SD_IDS="0.97"))
    assert len(lines) == 2, out[-3000:]
    _check_diff_lines(lines, 200000)


def test_deterministic_reference_leg_is_identical_in_every_mode():
    """The reference's DETERMINISTIC configuration -- one thread, no accelerator: one hit list per query, clumps ascending
    (burst.c:4344-4476) -- on the bench workload's family database cut to a size its exhaustive path finishes in a minute or two
    (1 800 references in families of 30, ~340 clumps; 14 us of one core per query and clump): 2 400 reads of 100 symbols on both
    strands in the three modes whose output depends on the hit-list order, and configs[4]'s shape (800 reads of 320 symbols with IUPAC
    codes, FORAGE at 95 %).  Where the multi-threaded legs of this file may differ in thread-order-dependent lines (bounded and
    explained), this one must be IDENTICAL byte for byte: CAPITALIST votes and ties, the ALLPATHS / FORAGE duplicate hunt, the strand merge."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst12")):
        pytest.skip("compiled reference not present")
    from burst_amd import host
    jobs = []
    for read_len, thres, modes, iupac, edits, n_reads in ((100, 0.97, "CAPITALIST ALLPATHS FORAGE", 0.0, [0, 1, 2, 3], 2400), (320, 0.95, "FORAGE", 0.01, [0, 2, 4, 8, 12], 800)):
        work, refs, edx, acx = _bench_db(read_len, thres, n_base=60, n_variants=30, want_acx=False)
        reads = os.path.join(work, "det_reads_l%d.fa" % read_len)
        if not os.path.exists(reads):
            host.synth_reads(refs, reads, n_reads, read_len, edits, rc=True, iupac=iupac, seed=91)
        for mode in modes.split():      # (the four one-thread reference runs side by side: 131 s one after the other in round 5's suite)
            jobs.append((n_reads, dict(BURST_BENCH_DIR=work, SD_EDX=edx, SD_READS=reads, SD_MODES=mode, SD_IDS=str(thres), SD_EXTRA="-fr", SD_EXHAUSTIVE="1", SD_THREADS="1")))
    for lines, out in _scale_diffs_side_by_side(jobs):
        assert len(lines) == 1, out[-3000:]
        assert "IDENTICAL" in lines[0], lines[0]


def test_configs2_twelve_million_292bp_reads_allpaths():
    """BASELINE configs[2] at its defining size: 12 M 292-bp amplicon-like reads vs the GG97-like database, -m ALLPATHS -i 0.97,
    through the product's batch scheduler (bh_align_ranges: 2 M-read batches, staged one ahead).  Size-independent properties:
    full sensitivity (every read carries <= 8 edits <= its budget, so every unique query must be reported), every record within
    budget and at its query's minimum (ALLPATHS), the f32 identity of every record, positions inside the clump, records of a
    query contiguous and ordered; and the first 100 000 reads are diffed against the compiled reference."""
    work, refs, edx, acx = _bench_db(292, 0.97)
    from burst_amd import host
    n_reads = 12000000
    reads = os.path.join(work, "cfg2_reads_12m_292.fa")
    if not os.path.exists(reads + ".done"):
        host.synth_reads(refs, reads, n_reads, 292, list(range(9)), rc=False, iupac=0.0, seed=77)
        open(reads + ".done", "w").write("ok")
    db = host.Db.read(edx, acx)
    qs = host.QuerySet(reads, 0.97, rc=False, accel=True, K=int(db.c.K))
    assert qs.n_reads == n_reads
    dev = db.open_device(0)
    qs.pin()
    run = host.align_ranges(dev, qs, [(0, qs.n_uniq)], "ALLPATHS", 1 << 21)
    h = run.hits
    assert int(run.c.nBatches) == (qs.n_uniq + (1 << 21) - 1) >> 21
    q = h["q"].astype(np.int64)
    emac = host._view(qs.c.emac, qs.n_entries, np.uint16)
    qoff = host._view(qs.c.qoff, qs.n_entries + 1, np.uint64).astype(np.int64)
    found = np.zeros(qs.n_uniq, bool)
    found[q] = True
    assert found.all(), "%d unique queries without a record" % (~found).sum()
    assert (h["ed"] <= emac[q]).all()
    best = np.full(qs.n_uniq, 255, np.int64)
    np.minimum.at(best, q, h["ed"])
    assert (h["ed"] == best[q]).all()                        # ALLPATHS: only the minimum of the query survives
    ln = (qoff[q + 1] - qoff[q]).astype(np.float32)
    want = (np.float32(1.0) - h["ed"].astype(np.float32) / (ln + h["gapQ"].astype(np.float32))).astype(np.float32)
    assert want.tobytes() == h["score"].astype(np.float32).tobytes()
    clump_len = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    assert (h["refIx"] < db.c.totR).all()
    assert (h["finalPos"] >= 1).all() and (h["finalPos"] <= clump_len[h["refIx"] >> 4]).all()
    key = (q << 32) | h["refIx"].astype(np.int64)
    assert (np.diff(key) > 0).all()                          # sorted by (query, reference), no duplicates
    n_rec = len(h)
    run.close()
    dev.close()
    assert n_rec >= qs.n_uniq
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst12")):
        lines, out = _scale_diff(100000, dict(BURST_BENCH_DIR=work, SD_EDX=edx, SD_READS=reads, SD_MODES="BEST ALLPATHS", SD_IDS="0.97"))
        assert len(lines) == 2, out[-3000:]
        _check_diff_lines(lines, 100000)


def test_bench_workload_db15_parity():
    """The bench's own workload against the compiled reference: bench.py's database generator (low-redundancy: random base
    sequences x 2 variants; here 800 000 bases = half the bench's default size, 2.2 Gbp -- the accelerator is built on the device
    in two slices of clumps), DB15, 100-bp reads with 0-2 edits, -m BEST -i 0.98.  oracle/_ref/burst15 reads the .acx written from
    the device-built tables; burst_hip builds its accelerator itself (-ad -k 15).  BEST must be identical line for line."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst15")):
        pytest.skip("compiled reference not present")
    work, refs, edx, acx = _bench_db(100, 0.98, K=15, n_base=800000, n_variants=2)
    from burst_amd import host
    reads = os.path.join(work, "parity15_reads_300k.fa")
    if not os.path.exists(reads + ".done"):
        host.synth_reads(refs, reads, 300000, 100, [0, 1, 2], rc=False, iupac=0.0, seed=42)
        open(reads + ".done", "w").write("ok")
    lines, out = _scale_diff(200000, dict(BURST_BENCH_DIR=work, SD_EDX=edx, SD_READS=reads, SD_MODES="BEST", SD_IDS="0.98", SD_REF=os.path.join(ROOT, "oracle", "_ref", "burst15"),
                                          SD_HIP_ACCEL="-ad -k 15"))
    assert len(lines) == 1 and "IDENTICAL" in lines[0], out[-3000:]
    n_lines = int(re.search(r"(\d+) reference lines", lines[0]).group(1))
    assert n_lines > 190000, lines[0]


def test_configs4_shape_full_size():
    """BASELINE configs[4]'s shape at a size one device call cycle covers: 1 M 320-bp reads with 1 % IUPAC codes, both strands (-fr),
    -m FORAGE -i 0.95 (every placement within budget, burst.c:4224; ambiguous words burst.c:3232-3236), through the product's batch
    scheduler.  Size-independent properties: every read's home placement is reported (a read carries <= 12 edits <= its budget of
    16), every record within budget, the f32 identity of every record, positions inside the clump, records ordered by (query,
    reference) without duplicates, both strands present; and the first 400 reads are diffed against the compiled reference (which needs most of a minute for them on 256 threads)
    (FORAGE and BEST; FORAGE under the "every differing line explained" rule of _check_diff_lines)."""
    work, refs, edx, acx = _bench_db(320, 0.95)
    from burst_amd import host
    n_reads = 1000000          # (2 M until round 6: the properties do not depend on the size, the suite's time does)
    reads = os.path.join(work, "cfg4_reads_1m_320_iupac_fr.fa")
    if not os.path.exists(reads + ".done"):
        host.synth_reads(refs, reads, n_reads, 320, [0, 2, 4, 8, 12], rc=True, iupac=0.01, seed=99)
        open(reads + ".done", "w").write("ok")
    db = host.Db.read(edx, acx)
    qs = host.QuerySet(reads, 0.95, rc=True, accel=True, K=int(db.c.K))
    assert qs.n_reads == n_reads and qs.n_entries == 2 * qs.n_uniq
    dev = db.open_device(0)
    qs.pin()
    run = host.align_ranges(dev, qs, [(0, qs.n_uniq)], "FORAGE", 1 << 19)
    h = run.hits
    q = h["q"].astype(np.int64)
    emac = host._view(qs.c.emac, qs.n_entries, np.uint16)
    qoff = host._view(qs.c.qoff, qs.n_entries + 1, np.uint64).astype(np.int64)
    six = q % qs.n_uniq                                     # entry -> unique query (forward entries first, then the reverse complements)
    found = np.zeros(qs.n_uniq, bool)
    found[six] = True
    assert found.all(), "%d unique queries without a record" % (~found).sum()
    assert (h["ed"] <= emac[q]).all()
    assert (h["rc"] == (q >= qs.n_uniq)).all() and h["rc"].any() and not h["rc"].all()
    ln = (qoff[q + 1] - qoff[q]).astype(np.float32)
    want = (np.float32(1.0) - h["ed"].astype(np.float32) / (ln + h["gapQ"].astype(np.float32))).astype(np.float32)
    assert want.tobytes() == h["score"].astype(np.float32).tobytes()
    clump_len = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    assert (h["refIx"] < db.c.totR).all()
    assert (h["finalPos"] >= 1).all() and (h["finalPos"] <= clump_len[h["refIx"] >> 4]).all()
    # records of an entry contiguous and ascending by reference, no duplicates; entries ascending inside a batch (a batch = its
    # forward entries, then their reverse complements)
    same = np.diff(q) == 0
    assert (np.diff(h["refIx"].astype(np.int64))[same] > 0).all()
    assert int((~same).sum()) + 1 == len(np.unique(q))
    assert len(h) > qs.n_uniq                                # FORAGE: more than one placement per query on a database of families
    run.close()
    dev.close()
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst12")):
        lines, out = _scale_diff(400, dict(BURST_BENCH_DIR=work, SD_EDX=edx, SD_READS=reads, SD_MODES="FORAGE BEST", SD_IDS="0.95", SD_EXTRA="-fr"))      # (800 reads until round 6: the reference needs 45 s per mode for them)
        assert len(lines) == 2, out[-3000:]
        # (which of two overlapping shears the reference prints depends on the order its 256 threads found them in: 12 .. 26 of ~1 280 lines
        # from run to run on the same inputs -- the count is bounded loosely, what is strict is that every one of them is explained)
        _check_diff_lines(lines, 400, frac=0.04)
