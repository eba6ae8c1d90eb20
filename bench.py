#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the BURST alignment hot path on MI355X, on the metric's own shape.

Workload (BASELINE.json configs[3] / `metric`, at the size one GPU box can build in a bench run): synthetic 100-bp reads
(0-2 edits, LLsim-style) against a RefSeq stand-in database with a DB15 (K = 15) accelerator, `-m BEST -i 0.98`.  The real
31.5 GB RefSeq subset is not reachable offline; the stand-in is a low-redundancy synthetic database (default 1.6 M random base
sequences x 2 variants at 5 % divergence = 3.2 M references, 4.5 Gbp: every 15-mer has ~4 unrelated list entries, as a
collision-dominated RefSeq-scale DB15 would have many more), its accelerator BUILT ON THE DEVICE from the .edx (no .acx is read
or uploaded).  The database is as large as the box holds (--db-scale auto: the metric's own 31.5 GB .edx = 11.37 units on a 288 GB device in a
300 GB container, with the compiled reference run on it beside the device path); `config.extrapolation.measured_sizes` carries the bench line
measured at three database sizes up to the metric's own (profiles/r06_sizes.json).

A step = one batch of reads through the WHOLE device path as the product runs it: bench.py calls the C batch scheduler of
the `burst_hip` command line (bh_align_ranges, burst_amd/csrc/host/bh_align.c) -- each step's batch is staged afresh from
host memory (copies + device-side routing on the staging stream, one batch ahead of the batch being aligned), aligned
(seeds -> prefilter -> two-stage bit-parallel edit distance -> re-scoring -> sorted records) and its records handed back to
host memory behind the call.  Staging is therefore INSIDE the timed region.  The steps cycle through --pool distinct
batches of the sorted unique queries.  N > 1 (one process per GPU under torch.distributed.run): the database is replicated and the
single-GPU job of --steps batches is cut into N equal shares of unique queries (strong scaling, the default: the fixed job north_star
asks about) through the product's multi-rank search (bh_search_multi_ex); the path partitions, so there is no collective on the
data path -- every rank's page-locked record buffer is a shared-memory segment rank 0 has mapped (bh_node.c), the hand-over inside
the timed region is one word per rank and rank 0 reads the records where they lie (`handover` in the JSON line).  Extra keys of the
N > 1 line: `weak_scaling` (every rank --steps batches of its own), `configs3_job` (10 M reads over the N GPUs), `rccl` (the same job
with bhip_comm_gather_hits across all N ranks inside the timed region: rccl_ranks, rccl_gather_ms).  torch.distributed only carries the
job name and the barriers around the timed regions.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel by time on the critical path (HIP events on the stream it runs on);
`roofline.per_kernel` lists the others with their own bound; PMC-derived traffic / VALU figures come from profiles/
(tools/profile_round.sh).  `cpu_baseline` is the compiled reference itself (oracle/_ref/burst15, one thread per core of the job's
CPU quota) on a bounded sample of the same reads and database (its .acx is a NAMED PIPE fed from the device-built tables while it reads
them -- read_accelerator, burst.c:3535-3594, only reads forward -- so the 167 GB accelerator of the metric's database never exists as a file):
one run, its alignment loops timed between its own progress lines ("Using ACCELERATOR to align" .. "Search complete", burst.c:4048-4525)
read line by line (pseudo-terminal, or `stdbuf -oL` on a pipe; without either, the differential wall time of two sample sizes).  Where the reference's accelerated run cannot fit the
job's memory (it holds the .edx and the .acx) the parity check runs its exhaustive path on a small sample.  `parity_vs_reference`: the .b6 the reference wrote for that sample against
the .b6 of the device path for the same reads (outside the timed region).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the HIP runtime multiplexes all streams of a process onto 4 hardware queues by default; the library needs 4 of its own
# (chain, prefilter, staging, record hand-over) next to whatever torch uses.  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# database sizes `--db-scale auto` chooses from: the metric's own (31.5 GB .edx), the largest of round 3's slope runs, ... the
# database of rounds 1-3 (2.77 GB)
AUTO_SCALES = (11.37, 7.0, 4.0, 2.0, 1.0)
# per unit of scale (measured, profiles/): 4.48 Gbp of references = 4.6 GB of FASTA, .edx 2.77 GB, 4.75 G accelerator entries
# (device: 4 B each; the reference's .acx file: 3 B each + a 4.3 GB length table), 4.9 GB of offset lines on the device whatever the size
UNIT_FASTA, UNIT_EDX, UNIT_ENTRIES = 4.6e9, 2.77e9, 4.75e9


def memory_limit():
    """bytes of host memory this job may use: the machine's, or the container's (cgroup v2 / v1) when that is less -- files in a
    RAM-backed work directory (/dev/shm) count against it"""
    try:
        lim = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        lim = 64 << 30
    for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(f).read().strip()
            if v.isdigit() and 0 < int(v) < lim:
                lim = int(v)
        except OSError:
            pass
    return lim


def memory_in_use():
    for f in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
        try:
            return int(open(f).read().strip())
        except (OSError, ValueError):
            pass
    return 0


def memory_peak_str():
    try:
        return "%.0f GB" % (int(open("/sys/fs/cgroup/memory.peak").read().strip()) / 1e9)
    except (OSError, ValueError):
        return "unknown"


def effective_cores():
    """host cores this job can really use: the visible ones, or the container's CPU quota (cgroup cpu.max) when that is less"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(round(int(q) / int(per)))))
    except (OSError, ValueError):
        pass
    return n


def sizes_at(sc):
    """(bytes of the .edx, of the device copy, of the .acx file the reference reads) at a database scale"""
    edx = sc * UNIT_EDX
    return edx, edx * 1.06 + sc * UNIT_ENTRIES * 4 + 4.9e9 + 14e9, sc * UNIT_ENTRIES * 3 + 4.3e9


def reference_memory(edx, acx_file, K):
    """bytes the compiled reference holds while it aligns: its copy of the .edx (read_edb, burst.c:2842-2975), the whole .acx -- length
    table and lists (read_accelerator, 3535-3594) --, one pointer per word (`Forest`, 8 B x 4^K) and its threads' scratch"""
    return edx + acx_file + 8.0 * 4 ** K + 8e9


def host_need(sc, with_reference, ram_backed, world=1, K=15):
    """host memory a run at this scale touches at its worst moment.  Building: one part (2.5 units: FASTA + ~5 bytes per base in the
    QUICK builder) beside the parts' .edx files and, at the end, the merged file.  Running: the .edx file and this process's copy.
    With the reference: what the reference holds (reference_memory) beside the .edx FILE -- its .acx is a named pipe fed from the device
    (AcxFeed), not a file, and this process lets go of its copy of the database meanwhile.  Files only count when the work directory is
    RAM-backed"""
    edx, _, acx = sizes_at(sc)
    part = min(sc, 2.5) * UNIT_FASTA
    files = (2 * edx + 1.5e9) if ram_backed else 0
    build = part * (6 if ram_backed else 5) + files
    run = world * (edx + 4e9) + files / 2 + 8e9          # (every rank's process holds its own copy of the database for the upload)
    ref = ((edx if ram_backed else 0) + reference_memory(edx, acx, K) + 10e9) if with_reference else 0
    return max(build, run, ref) + 6e9


def pick_setup(args, free_hbm, want_cpu_baseline, world=1):
    """(db-scale, work directory): the largest database of AUTO_SCALES this box holds -- on the device (references + 4-byte records +
    offset lines + batch buffers), in the host memory the job may use (see host_need; with 8 % of slack) and in the work directory
    (FASTA parts, .edx, reads; the .acx the compiled reference reads is a named pipe)"""
    import shutil
    def free_of(d):
        try:
            while not os.path.exists(d):
                d = os.path.dirname(d) or "/"
            return shutil.disk_usage(d).free
        except OSError:
            return 0
    ram = memory_limit() * 0.92
    try:      # what other jobs left in RAM-backed files counts against the same limit and cannot be reclaimed
        ram -= shutil.disk_usage("/dev/shm").used
    except OSError:
        pass
    dirs = [args.workdir] if args.workdir else ["/dev/shm/burst_amd_bench", "/tmp/burst_amd_bench"]
    scales = AUTO_SCALES if args.db_scale == "auto" else (float(args.db_scale),)
    for sc in scales:
        edx, dev_need, acx_file = sizes_at(sc)
        for with_ref in ((True, False) if args.db_scale != "auto" else (want_cpu_baseline,)):      # (auto: a size at which the reference runs beside it, if one is wanted)
            disk_need = min(sc, 2.5) * UNIT_FASTA + 2 * edx + 4e9
            for d in dirs:
                if dev_need <= free_hbm and disk_need <= free_of(d) and host_need(sc, with_ref, d.startswith("/dev/shm"), world, args.K) <= ram:
                    return sc, d
    if args.db_scale != "auto":      # an explicit size is taken at its word (the allocation says when it does not fit)
        return float(args.db_scale), dirs[0]
    return 0.5, dirs[-1]


# What FETCH_SIZE means on gfx950, measured on the prefilter's own access pattern (round 6; tools/calibrate_fetch.sh runs
# tools/ubench/run_gather -- runs of N consecutive dwords at random 4-byte-aligned addresses, 64 adjacent lanes on 64 adjacent
# positions, every byte and every line known to the host -- under rocprofv3 --pmc FETCH_SIZE).
TRAFFIC_CALIBRATION = {
    "source": "profiles/r06e_fetch_calibration_runs.txt (tools/calibrate_fetch.sh)",
    "finding": "FETCH_SIZE x 1024 = 64 B x the number of DISTINCT 128-BYTE LINES a dispatch requests, whatever the pattern: ratio 0.500 / 0.500 / 0.503 / 0.503 "
               "against the known line count for runs of 1 / 16 / 40 / 164 dwords (single dwords: = 1.000 x 64-byte sectors, which is what profiles/r02k_fetch_calibration.txt "
               "saw and rounds 2-5 took for 'the counter is exact for gathers')",
    "reading": "`traffic` (2 x FETCH_SIZE + WRITE_SIZE) = the bytes of every requested line = what moved if lines move whole; `traffic_half_lines` (FETCH_SIZE + WRITE_SIZE) = one "
               "64-byte half per requested line, what moved if a line of which one half is wanted moves that half only.  The counter cannot tell the two apart; a list of "
               "~41 four-byte records at a random address spans 2.25 lines but 3.5 64-byte sectors, so the truth for k_prefilter_cq lies between the two and both are above "
               "the device-record bytes only in the first reading",
}


def shape_name(args):
    """which BASELINE.json configuration the run's shape is (read length, mode, identity, strands): the line's workload says so"""
    key = (args.read_len, args.mode, float(args.id), bool(args.fr))
    return {(100, "BEST", 0.98, False): "BASELINE configs[3] shape", (100, "CAPITALIST", 0.97, False): "BASELINE configs[1] shape",
            (292, "ALLPATHS", 0.97, False): "BASELINE configs[2] shape", (320, "FORAGE", 0.95, True): "BASELINE configs[4] shape"}.get(key, "a shape of its own")


def db_paths(workdir, args):
    args.db_qlen = args.read_len + max(10, args.read_len // 10)
    tag = "b%d_v%d_l%d_q%d_i%s_k%d%s" % (args.n_base, args.n_variants, args.ref_len, args.db_qlen, args.id, args.K, "" if getattr(args, "db_profile", "pairs") == "pairs" else "_" + args.db_profile)
    return (os.path.join(workdir, "refs_%s.fa" % tag), os.path.join(workdir, "db_%s.edx" % tag), os.path.join(workdir, "db_%s.acx" % tag))


def reads_path(workdir, args):
    edits = [int(x) for x in args.edits.split(",")]
    return os.path.join(workdir, "reads_%d_l%d_e%s_u%s_f%d%s.fa" % (args.reads * args.pool, args.read_len, "-".join(map(str, edits)), args.iupac, int(args.fr),
                                                                    "" if getattr(args, "db_profile", "pairs") == "pairs" else "_" + args.db_profile))


# A database is BUILT in parts of at most this many base sequences (x variants; 2.5 units of scale = 11 Gbp of FASTA): the QUICK
# builder holds about five bytes per base while it sorts the fragments, and the GPU boxes give a job 300 GB.  A part is what
# `-d QUICK` makes of its share of the references; the parts are laid end to end (bh_edx_merge).  Up to 2.5 units the database is
# one part -- the same file as in rounds 1-3.
PART_BASES = 4000000


def db_parts(args):
    """the parts a database is built in: (first base sequence, base sequences, variants per base sequence, divergence, seed).  Default
    profile: every base sequence with --n-variants variants at --variant-rate (low redundancy: what the headline is quoted on).  Profile
    `strains`: 70 % of the content as that, 30 % in families of near-identical sequences -- 60 / 200 / 500 variants at 1 / 0.5 / 0.1 %
    divergence, a tenth of the content each -- the strain-level redundancy of complete-genome collections: a read from such a family has
    dozens to hundreds of candidate references"""
    def cut(first, n, nv, rate, seed):
        per = max(1, PART_BASES * 2 // nv)          # (a part holds about PART_BASES x 2 sequences)
        k = max(1, -(-n // per))
        size = -(-n // k)
        size += (-size) % 8                         # (x variants x 3 fragments: every part but the last fills its last clump)
        return [(first + i * size, min(n, (i + 1) * size) - i * size, nv, rate, seed) for i in range(k) if i * size < n]
    if getattr(args, "db_profile", "pairs") != "strains":
        return cut(0, args.n_base, args.n_variants, args.variant_rate, 7)
    seqs = args.n_base * args.n_variants
    parts = cut(0, int(args.n_base * 0.7), args.n_variants, args.variant_rate, 7)
    first = args.n_base
    for nv, rate, seed in ((60, 0.01, 11), (200, 0.005, 12), (500, 0.001, 13)):
        n = max(1, int(seqs * 0.1 / nv))
        parts += cut(first, n, nv, rate, seed)
        first += n + 8
    return parts


def build_db(workdir, args, rank=0, reads_fa=None):
    """rank 0 writes the shared database (args: read_len, n_base, n_variants, ref_len, variant_rate, id, K, db_profile) -- and, with reads_fa,
    the read pool drawn from its references (part by part: the reference FASTA of a part is deleted once its clumps and reads exist, unless
    the database is a single part and args.drop_refs is off); returns (refs, edx, acx, done marker)"""
    from burst_amd import host
    os.makedirs(workdir, exist_ok=True)
    refs, edx, acx = db_paths(workdir, args)
    done = edx + ".done"
    if rank == 0 and not (os.path.exists(done) and (reads_fa is None or os.path.exists(reads_fa + ".done"))):
        t = time.time()
        edits = [int(x) for x in args.edits.split(",")] if reads_fa else []
        specs = db_parts(args)
        n_parts = len(specs)
        tot_seqs = sum(n * nv for _, n, nv, _, _ in specs)
        n_pool = args.reads * args.pool if reads_fa else 0
        t_ref = t_cl = t_rd = 0.0
        parts, first_read = [], 0
        for p, (b0, nb, nv, rate, seed) in enumerate(specs):
            fa = refs if n_parts == 1 else refs + ".part%d" % p
            ex = edx if n_parts == 1 else edx + ".part%d" % p
            t0 = time.time()
            host.synth_refs(fa, nb, nv, args.ref_len, rate, seed, first_base=b0)
            t1 = time.time()
            if not os.path.exists(done):
                db = host.Db.from_fasta(fa, args.db_qlen, args.id, shear_len=500)
                db.write(ex, None, db_qlen=args.db_qlen, thres=args.id)
                db.close()
            t2 = time.time()
            if reads_fa and not os.path.exists(reads_fa + ".done"):
                n_here = n_pool * (nb * nv) // tot_seqs if p + 1 < n_parts else n_pool - first_read      # (reads in proportion to the part's share of the content)
                host.synth_reads(fa, reads_fa, n_here, args.read_len, edits, rc=args.fr, iupac=args.iupac, seed=42 + p, first_read=first_read, append=p > 0)
                first_read += n_here
            t3 = time.time()
            if n_parts > 1 or getattr(args, "drop_refs", False):
                os.remove(fa)      # (as large as the database in bases)
            parts.append(ex)
            t_ref += t1 - t0; t_cl += t2 - t1; t_rd += t3 - t2
        t4 = time.time()
        if n_parts > 1 and not os.path.exists(done):
            host.edx_merge(parts, edx)
            for ex in parts:
                os.remove(ex)
        open(done, "w").write("ok")
        if reads_fa:
            open(reads_fa + ".done", "w").write("ok")
        log("[bench] database built in %.1f s in %d part(s) (references %.1f s, clumps + .edx %.1f s, reads %.1f s, merge %.1f s); the accelerator is built on the device"
            % (time.time() - t, len(parts), t_ref, t_cl, t_rd, time.time() - t4))
    return refs, edx, acx, done


def ensure_acx(edx, acx, K):
    """an .acx FILE for the database (the compiled reference reads one; so does burst_hip -a): `burst_hip --make-acx`, which builds the
    accelerator on the device when there is one"""
    if not os.path.exists(acx + ".done"):
        t = time.time()
        subprocess.check_call([os.path.join(ROOT, "burst_amd", "burst_hip"), "-r", edx, "--make-acx", acx, "-k", str(K)], stdout=subprocess.DEVNULL)
        open(acx + ".done", "w").write("ok")
        log("[bench] %s written in %.1f s" % (acx, time.time() - t))
    return acx


def build_inputs(workdir, args, rank):
    """rank 0 writes the shared database and the shared read pool"""
    os.makedirs(workdir, exist_ok=True)
    db_paths(workdir, args)          # (sets args.db_qlen)
    reads_fa = reads_path(workdir, args)
    refs, edx, acx, done = build_db(workdir, args, rank, reads_fa)
    return refs, edx, acx, reads_fa, done


class AcxFeed:
    """The reference's .acx as a NAMED PIPE fed from the tables the device built.  read_accelerator (burst.c:3535-3594) is fopen + fgetc +
    four sequential fread calls and never seeks, so the file it is given can be a FIFO: bh_acx_write_from_device writes the header, the
    length table, the lists (run by run, fetched from the device and packed as they go) and the BadList into it while the reference
    reads them into its own memory -- the 167 GB accelerator of the metric's database never exists as a file, which is what lets the
    reference's accelerated run fit the job's memory beside the .edx.  One feed per reference run (start before the run, finish after)."""

    def __init__(self, dev, n_clumps, K, path):
        self.dev, self.n_clumps, self.K, self.path = dev, n_clumps, K, path
        self.thread = self.error = None
        self.seconds = 0.0

    def start(self):
        import threading
        from burst_amd import host
        try:
            os.remove(self.path)
        except OSError:
            pass
        os.mkfifo(self.path)
        stub = host.Db()                      # (the writer wants the clump count -- the list format -- and nothing else of the database)
        stub.c.numRclumps = self.n_clumps
        def work():
            t = time.time()
            try:
                stub.acx_write_from_device(self.dev, self.K, self.path)      # (blocks in fopen until the reference opens its end)
            except Exception as e:
                self.error = str(e)
            self.seconds = time.time() - t
        self.error = None
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def finish(self):
        """after the reference has ended: a writer still waiting for a reader (the reference died before it opened the pipe) is released"""
        if self.thread is not None and self.thread.is_alive():
            try:
                fd = os.open(self.path, os.O_RDONLY | os.O_NONBLOCK)
                time.sleep(0.2)
                os.close(fd)          # (the writer's next write fails with EPIPE; Python ignores SIGPIPE)
            except OSError:
                pass
            self.thread.join(60)
        self.thread = None
        try:
            os.remove(self.path)
        except OSError:
            pass
        return self.error


def run_reference_timed(cmd, feed=None):
    """run the compiled reference with its stdout line-buffered -- on a pseudo-terminal, or (no pty devices in the container) through a
    pipe under `stdbuf -oL` -- and stamp its own progress lines: returns (return code, whole wall time, seconds between its "Using
    ACCELERATOR to align ..." (burst.c:4048; without -a: "Searching best paths ...", 4325) line and its "Search complete" line (4525) --
    the alignment loops of do_alignments -- or None when the lines were not seen or could not be stamped, the output's tail)"""
    import select
    import shutil
    master = slave = sb = None
    try:
        import pty
        master, slave = pty.openpty()
    except Exception:
        master = slave = None
    if feed is not None:
        feed.start()
    t0 = time.time()
    if master is not None:
        p = subprocess.Popen(cmd, stdout=slave, stderr=slave, close_fds=True)
        os.close(slave)
        fd = master
    else:
        sb = shutil.which("stdbuf")
        p = subprocess.Popen(([sb, "-oL", "-eL"] if sb else []) + cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, close_fds=True)
        fd = p.stdout.fileno()
    buf, t_search, t_done, tail = b"", None, None, []
    while True:
        try:
            r, _, _ = select.select([fd], [], [], 0.5)
            if r:
                chunk = os.read(fd, 65536)
                if not chunk:
                    break
                now = time.time()
                buf += chunk
                while b"\n" in buf:
                    line, buf = buf.split(b"\n", 1)
                    tail.append(line.decode("utf-8", "replace").strip())
                    if t_search is None and (b"Using ACCELERATOR" in line or b"Searching best paths" in line):
                        t_search = now
                    elif t_search is not None and t_done is None and b"Search complete" in line:
                        t_done = now
            elif p.poll() is not None:
                break
        except OSError:      # (the child closed its side)
            break
    rc = p.wait()
    wall = time.time() - t0
    if master is not None:
        os.close(master)
    if feed is not None:
        err = feed.finish()
        if err:
            tail.append("[.acx feed] " + err)
    align = (t_done - t_search) if (t_search is not None and t_done is not None) else None
    if master is None and not sb:      # a fully buffered pipe delivers the lines together at the end: not a measurement
        align = None
    return rc, wall, align, "\n".join(tail[-12:])


def cpu_baseline(edx, acx, reads_fa, args, feed=None):
    """the compiled reference on the host cores, ONE run on the sample (feed: its .acx is the named pipe `acx`, fed from the device): its align phase is the time between its own "Searching best
    paths ..." and "Search complete" lines (the OpenMP loops this repository replaces: scour + aded_mat16 + reScoreM + hit capture),
    stamped as they arrive on a pseudo-terminal.  (Rounds 1-3 took the difference of two runs' wall times, which at 40 s of database
    load per run had a spread as large as the 2-4 s it was after.)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "burst%d" % args.K)
    if args.no_cpu_baseline or not os.path.exists(exe):
        return None
    cores = effective_cores()
    n = args.cpu_sample
    sample = os.path.join(os.path.dirname(reads_fa), "cpu_sample_%d.fa" % n)
    with open(reads_fa, "rb") as f, open(sample, "wb") as o:
        for _ in range(2 * n):
            o.write(f.readline())
    rc, wall, align_s, tail = run_reference_timed([exe, "-r", edx, "-a", acx, "-q", sample, "-o", sample + ".b6", "-m", args.mode, "-i", str(args.id),
                                                   "-t", str(cores)] + (["-fr"] if args.fr else []), feed)
    cpu_baseline.wall, cpu_baseline.feed_s = wall, (feed.seconds if feed is not None else None)
    if rc != 0:
        log("[bench] reference failed:", tail[-400:])
        return None
    cpu_baseline.sample_fa, cpu_baseline.sample_b6 = sample, sample + ".b6"        # parity_vs_reference compares with it
    if align_s is None:
        # the progress lines could not be stamped (no pty, no stdbuf): the difference of two runs' wall times, as in rounds 1-3
        n1 = max(1, n // 6)
        small = os.path.join(os.path.dirname(reads_fa), "cpu_sample_%d.fa" % n1)
        with open(reads_fa, "rb") as f, open(small, "wb") as o:
            for _ in range(2 * n1):
                o.write(f.readline())
        if feed is not None:
            feed.start()
        t_ = time.time()
        r_ = subprocess.run([exe, "-r", edx, "-a", acx, "-q", small, "-o", small + ".b6", "-m", args.mode, "-i", str(args.id), "-t", str(cores), "--noprogress"] + (["-fr"] if args.fr else []),
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        w1 = time.time() - t_
        if feed is not None:
            feed.finish()
        dt = wall - w1
        if r_.returncode != 0 or dt < 0.02 * wall or dt < 0.5:
            cpu_baseline.too_small = "the reference's wall time for %d and %d reads (%.2f s, %.2f s: database load) does not resolve its align phase" % (n1, n, w1, wall)
            return None
        return {"value": (n - n1) / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "oracle/_ref/burst%d -t %d, same .edx/.acx, -m %s -i %s; differential wall time of the first %d vs %d reads of the pool (%.2f s vs %.2f s), database load cancelled "
                          "(its progress lines could not be stamped here)" % (args.K, cores, args.mode, args.id, n1, n, w1, wall)}
    if align_s < 0.2:
        cpu_baseline.too_small = "the reference's align phase on %d reads was not resolvable (%.3f s of %.1f s wall)" % (n, align_s, wall)
        return None
    return {"value": n / align_s, "unit": "reads/s", "cores": cores, "kind": "reference",
            "sample": "oracle/_ref/burst%d (reference compiled with gcc -O3 -march=x86-64-v3 -fopenmp) -t %d (%d hardware threads visible%s), same .edx/.acx, "
                      "-m %s -i %s, the first %d reads of the pool: %.2f s between its own 'Using ACCELERATOR to align' and 'Search complete' lines (its alignment loops, "
                      "burst.c:4048-4525; whole run %.1f s, most of it the database load)"
                      % (args.K, cores, os.cpu_count() or 1, "; the job's CPU quota is %d" % cores if cores < (os.cpu_count() or 1) else "", args.mode, args.id, n, align_s, wall)}


def exhaustive_reference_sample(edx, reads_fa, args, n_clumps):
    """When the reference's accelerated run does not fit this box (it holds the .edx and the .acx in memory next to the .acx FILE), its
    exhaustive path (no -a: every query against every clump, burst.c:4320-4488) gives the same hits by construction -- on a sample
    sized for about two minutes of the host cores (14 us of one core per query and clump, measured).  Sets the sample for
    parity_vs_reference; returns a description"""
    exe = os.path.join(ROOT, "oracle", "_ref", "burst%d" % args.K)
    cores = effective_cores()
    n = int(max(8, min(2000, 120.0 * cores / (max(1, n_clumps) * 14e-6))))
    sample = os.path.join(os.path.dirname(reads_fa), "ref_exhaustive_%d.fa" % n)
    with open(reads_fa, "rb") as f, open(sample, "wb") as o:
        for _ in range(2 * n):
            o.write(f.readline())
    t = time.time()
    r = subprocess.run([exe, "-r", edx, "-q", sample, "-o", sample + ".b6", "-m", args.mode, "-i", str(args.id), "-t", str(cores), "--noprogress"] + (["-fr"] if args.fr else []),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        log("[bench] reference (exhaustive) failed:", r.stdout[-400:])
        return None
    cpu_baseline.sample_fa, cpu_baseline.sample_b6 = sample, sample + ".b6"
    return {"reads": n, "seconds": time.time() - t, "cores": cores,
            "what": "oracle/_ref/burst%d WITHOUT its accelerator (exhaustive: every query against every clump) on the first %d reads of the pool, -t %d: the accelerated run "
                    "needs the .edx and the .acx in memory beside the .acx file, which this box's memory limit does not allow at this database size" % (args.K, n, cores)}


def end_to_end(edx, reads_fa, args, device):
    """the whole command line on the read pool: burst_hip -r DB.edx -ad -q reads -o out.b6 (database read, accelerator built on the
    device, query ingest beside it, search, consolidation, .b6 written) -- its own 'Alignment time' and phase lines"""
    import re
    out = reads_fa + ".e2e.b6"
    cmd = [os.path.join(ROOT, "burst_amd", "burst_hip"), "-r", edx, "-ad", "-k", str(args.K), "-q", reads_fa, "-o", out, "-m", args.mode, "-i", str(args.id), "--device", str(device)] + (["-fr"] if args.fr else [])
    best, first = None, None
    # Two runs.  The first follows whatever held the device a moment ago (this process's own handle: 255 GB at the metric's size) and pays for
    # it -- the memory of the process before is not back at once, its first large allocations wait seconds; the second
    # starts after a pause on a quiet device with the files in the page cache, as the bench's own inputs are: that one is reported, the
    # first beside it as `right_behind_another_process_s`.
    settle = 20.0 if os.path.getsize(edx) > 8e9 else 0.0
    failed = []          # a run that did not end with its "Alignment time" line: exit code and the end of its output are kept, and one more run is made
    it = 0
    while it < 2:
        if it and settle:
            time.sleep(settle)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        m = re.search(r"Alignment time: ([\d.]+) seconds", r.stdout)
        if r.returncode != 0 or not m:
            failed.append({"returncode": r.returncode, "after_runs": it, "output_tail": r.stdout[-1500:]})
            try:
                open(out + ".failed%d.log" % len(failed), "w").write(r.stdout)
            except OSError:
                pass
            if len(failed) > 1:
                return {"error": failed[-1]["output_tail"][-400:], "failed_runs": failed}
            time.sleep(5.0)
            continue
        n = int(re.search(r"Parsed (\d+) queries", r.stdout).group(1))
        best = {"reads": n, "seconds": float(m.group(1)), "reads_per_s": n / float(m.group(1)), "lines": int(re.search(r"Wrote (\d+) alignments", r.stdout).group(1)),
                "phases_s": {k.strip(): float(v) for k, v in re.findall(r"\[([a-z ,()+.]+?)\s+([\d.]+) s", r.stdout)},
                "command": "burst_hip -r DB.edx -ad -k %d -q <%d reads> -o out.b6 -m %s -i %s (second of two runs%s)" % (args.K, n, args.mode, args.id, ", %.0f s after the first" % settle if settle else "")}
        if it == 0:
            first = best["seconds"]
        it += 1
    if best is not None:
        best["right_behind_another_process_s"] = first
        if failed:
            best["failed_runs"] = failed
    try:
        os.remove(out)
    except OSError:
        pass
    return best


def share_of_job(ranges, rank, world):
    """the rank-th of `world` equal parts of a job given as ranges of unique queries aligned one after the other: the ranges (or
    pieces of them) that make up [rank, rank + 1) / world of the job's total"""
    if world == 1:
        return list(ranges)
    total = sum(b - a for a, b in ranges)
    lo, hi, out, pos = total * rank // world, total * (rank + 1) // world, [], 0
    for a, b in ranges:
        x0, x1 = max(lo, pos), min(hi, pos + (b - a))
        if x1 > x0:
            out.append((a + x0 - pos, a + x1 - pos))
        pos += b - a
    return out


def parity_vs_reference(dev, db, args, batch_uniq):
    """the reference's .b6 for the cpu_baseline sample against the device path's for the same reads (same database, same flags)"""
    from burst_amd import host
    sample, ref_b6 = getattr(cpu_baseline, "sample_fa", None), getattr(cpu_baseline, "sample_b6", None)
    if not sample or not os.path.exists(ref_b6):
        return None
    qs = host.QuerySet(sample, args.id, rc=args.fr, accel=True, K=args.K)
    run = host.align_ranges(dev, qs, [(0, qs.n_uniq)], args.mode, batch_uniq)
    out = sample + ".hip.b6"
    host.report(out, db, qs, run.hits, args.mode, 0)
    a = sorted(open(ref_b6, "rb").read().splitlines())
    b = sorted(open(out, "rb").read().splitlines())
    sa, sb = set(a), set(b)
    explained = None
    if a != b and args.mode != "BEST":
        # the modes that print several placements per query keep, of two overlapping ones, whichever the reference's threads met first
        # (DUPE_HUNT, burst.c:4563-4570; equally voted references in CAPITALIST, 4763-4776): a differing reference line is explained
        # when it is one of the placements the device path computes for that query (printed without the duplicate hunt)
        nd = sample + ".hip.nd.b6"
        host.report(nd, db, qs, run.hits, "ALLPATHS" if args.mode == "CAPITALIST" else args.mode, host.REP_NO_DUPE_HUNT)
        allp = set(open(nd, "rb").read().splitlines())
        explained = {"reference_lines_not_computed_by_the_device": len([x for x in sa - sb if x not in allp]), "same_line_count": len(a) == len(b),
                     "same_queries": sorted({x.split(b"\t")[0] for x in a}) == sorted({x.split(b"\t")[0] for x in b})}
    res = {"reads": qs.n_reads, "lines_reference": len(a), "lines": len(b), "identical": a == b, "only_reference": len(sa - sb), "only_device": len(sb - sa), "order_dependent_lines_explained": explained,
           "what": "sorted .b6 of oracle/_ref/burst%d vs the device path (bh_align_ranges + bh_report) on the first %d reads of the pool, -m %s -i %s" % (args.K, qs.n_reads, args.mode, args.id)}
    run.close(); qs.close()
    return res


def pmc_table(kernel_names):
    """(per-kernel PMC figures of the last profiling pass kept under profiles/ (tools/profile_round.sh), where they come from).  The
    table is a measurement of an EARLIER run of the same kernels, not of this one: it carries the commit and the command it was made
    with (`_meta`), and it is refused -- not silently used -- when it was made before the 4-byte accelerator records or does not
    know the kernels this run timed"""
    f = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        t = json.load(open(f))
    except Exception:
        return {}, "none (profiles/pmc_summary.json missing)"
    meta = t.pop("_meta", None)
    if not meta or meta.get("record_bytes") != 4:
        return {}, "profiles/pmc_summary.json refused: made before this round's kernels (no _meta / other record format)"
    missing = [k for k in kernel_names if k.split("<")[0].replace("_*", "_reg") not in t]
    if missing:
        return {}, "profiles/pmc_summary.json @ %s refused: it does not know %s" % (meta.get("commit"), ", ".join(missing))
    return t, "profiles/pmc_summary.json = %s @ commit %s (%s)" % (meta.get("tag"), meta.get("commit"), meta.get("command"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=2000000, help="reads per step (per rank with --scaling weak, all GPUs together with --scaling strong)")
    ap.add_argument("--pool", type=int, default=4, help="distinct batches the steps cycle through")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--db-scale", default="auto", help="multiplies --n-base (1 = 3.2 M references / 4.5 Gbp / .edx 2.77 GB; 11.37 = the metric's 31.5 GB .edx); "
                    "auto (default) = the largest of %s that the device, the host memory and the work directory hold" % (AUTO_SCALES,))
    ap.add_argument("--n-base", type=int, default=1600000, help="random base sequences (round 2's family-dominated database: --n-base 33000 --n-variants 30)")
    ap.add_argument("--n-variants", type=int, default=2)
    ap.add_argument("--db-profile", default="pairs", choices=["pairs", "strains"], help="pairs (default, what the headline is quoted on): every base sequence with --n-variants variants at --variant-rate; "
                    "strains: 30 %% of the content in families of 60 / 200 / 500 near-identical sequences (1 / 0.5 / 0.1 %% divergence), the rest as pairs -- the redundancy of complete-genome collections")
    ap.add_argument("--ref-len", type=int, default=1400)
    ap.add_argument("--variant-rate", type=float, default=0.05)
    ap.add_argument("--id", type=float, default=0.98)
    ap.add_argument("--K", type=int, default=15, choices=[12, 15], help="accelerator word length (the reference's DB12 / DB15 builds)")
    ap.add_argument("--mode", default="BEST")
    ap.add_argument("--workdir", default=os.environ.get("BURST_BENCH_DIR"), help="where the database, the reads and the reference's .acx are written (default: /dev/shm/burst_amd_bench when "
                    "/dev/shm has the room a large database needs, else /tmp/burst_amd_bench)")
    ap.add_argument("--no-continuity", action="store_true", help="skip the extra run on the small database of rounds 1-3 (key `continuity_small_db`)")
    ap.add_argument("--no-strains", action="store_true", help="skip the extra run on a database with strain-level redundancy (key `strains`: --db-profile strains at --strains-scale, with the reference beside it)")
    ap.add_argument("--strains-scale", default="1", help="database size of the `strains` leg in units of --db-scale (default 1 = 2.6 GB of .edx: the leg must fit the default run's few minutes; "
                    "the regime -- lane tasks and records per read -- does not depend on the size, the prefilter's share does: profiles/ keeps a 30 GB run)")
    ap.add_argument("--keep-files", action="store_true", help="leave the large files (reference FASTA, .acx) in the work directory")
    ap.add_argument("--cpu-sample", type=int, default=600000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the burst_hip command-line run on the read pool (end_to_end_reads_per_s)")
    ap.add_argument("--fr", action="store_true", help="also search reverse complements (-fr)")
    ap.add_argument("--iupac", type=float, default=0.0, help="fraction of read bases replaced by a compatible IUPAC code")
    ap.add_argument("--edits", default="0,1,2", help="edit counts sampled per read")
    ap.add_argument("--opt", action="append", default=[], help="library tuning option name=value (bhip_set_option), repeatable")
    ap.add_argument("--no-pin", action="store_true", help="leave the query arrays pageable")
    ap.add_argument("--drop-refs", action="store_true", help="delete the reference FASTA once the reads and the .edx exist (disk space of very large databases)")
    ap.add_argument("--no-prime", action="store_true", help="skip bhip_reserve and the priming call (profiling: every dispatch of the run is then a full-size batch)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"], help="N > 1: strong (default) = the single-GPU job of --steps batches cut into N equal shares of unique queries (the fixed job "
                    "north_star asks about); weak = every rank aligns --steps batches of --reads reads of its own (the job grows with N).  The line carries the other one as an extra key")
    ap.add_argument("--gather", default="shm", choices=["shm", "rccl"], help="N > 1: how the ranks' records reach rank 0 -- shared-memory segments rank 0 maps (default; no collective) or the library's RCCL gather")
    ap.add_argument("--acx-file", action="store_true", help="round 2's path: the accelerator from an .acx file (built by the host builder) instead of the device build")
    ap.add_argument("--no-short-job", action="store_true", help="skip the 1.25 M-read job (key `one_rank_share_of_configs3`)")
    ap.add_argument("--ab-host", action="store_true", help="N = 1: also time the job with the scheduler's staging between the batches instead of inside them (key `ab_host`)")
    ap.add_argument("--ab", action="append", default=[], help="N = 1: after the timed region, time the same steps again with these library options (name=value[,name=value]; repeatable) "
                    "on the same resident database -- extra key `ab` of the JSON line, an A/B on one box in one process")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    # BURST_BENCH_DIST1=1 (test hook, under torch.distributed.run with one process): take the N > 1 code path on a single GPU
    use_dist = world > 1 or (os.environ.get("BURST_BENCH_DIST1") == "1" and "MASTER_ADDR" in os.environ)
    # BURST_BENCH_DEVICE=<d> (test hook): every rank on device d -- the N > 1 command line of the driver on a one-GPU box.  RCCL
    # refuses two ranks on one device, so the launcher's plumbing (barriers, the max over ranks) is gloo then and the records take
    # the shared-memory hand-over, which needs no communicator
    one_dev = os.environ.get("BURST_BENCH_DEVICE")
    if one_dev is not None:
        local_rank = int(one_dev)
        if args.gather == "rccl" and world > 1:
            raise SystemExit("BURST_BENCH_DEVICE puts every rank on one device: --gather shm only")
    pdev = "cpu" if one_dev is not None else "cuda"          # where the plumbing's tensors live
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        import datetime
        long_wait = datetime.timedelta(hours=2)          # (the other ranks sit in a barrier while rank 0 writes a large database)
        if one_dev is not None:
            dist.init_process_group("gloo", timeout=long_wait)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=long_wait)

    from burst_amd import capi, host
    # the size of the database and where its files go: rank 0 decides (from its device's free memory, the host's memory and the
    # room in the work directory), every rank takes its word
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "burst%d" % args.K)
    want_base = world == 1 and not args.no_cpu_baseline and os.path.exists(ref_exe)
    try:
        free_hbm = torch.cuda.mem_get_info(local_rank)[0]
    except Exception:
        free_hbm = 0
    setup = [pick_setup(args, free_hbm, want_base, world)]
    if use_dist:
        dist.broadcast_object_list(setup, src=0)
    args.db_scale, args.workdir = float(setup[0][0]), setup[0][1]
    args.n_base = int(round(args.n_base * args.db_scale))
    args.drop_refs = args.drop_refs or (args.db_scale >= 2 and not args.keep_files)
    if rank == 0:
        log("[bench] database scale %.2f (.edx ~%.1f GB), work directory %s" % (args.db_scale, args.db_scale * UNIT_EDX / 1e9, args.workdir))
    refs, edx, acx, reads_fa, done = build_inputs(args.workdir, args, rank)
    os.sync()          # the files just written go to disk now, not beside the timed region (their write-back shares the PCIe root and the memory bus)
    if use_dist:
        dist.barrier()
    while not (os.path.exists(done) and os.path.exists(reads_fa + ".done")):
        time.sleep(0.2)

    t = time.time()
    if args.acx_file and not os.path.exists(acx):
        if rank == 0:
            dbb = host.Db.read(edx)
            host._chk(host.lib().bh_acx_build(C.byref(dbb.c), args.K, 1))
            host._chk(host.lib().bh_acx_write(C.byref(dbb.c), acx.encode()))
            dbb.close()
        if use_dist:
            dist.barrier()
    db = host.Db.read(edx, acx if args.acx_file else None, K=args.K)
    t_db = time.time() - t
    t = time.time()
    host.lib().bh_queries_sort_device(local_rank)          # large query files are sorted on this rank's own device
    qs = host.QuerySet(reads_fa, args.id, rc=args.fr, accel=True, K=args.K)
    t_q = time.time() - t
    t = time.time()
    if one_dev is not None and use_dist:      # ranks sharing the device build one after the other (the build sizes its scratch from what is free when it starts)
        for r in range(rank):
            dist.barrier()
    dev = db.open_device(local_rank, build_K=0 if args.acx_file else args.K)      # references up (lane-major layout); accelerator built on the device
    if one_dev is not None and use_dist:
        for r in range(rank, world):
            dist.barrier()
    t_dev = time.time() - t
    for kv in args.opt:
        name, _, val = kv.partition("=")
        dev.set_option(name, int(val))
    if not args.no_pin:
        qs.pin()
    info = dev.info()
    edx_bytes = os.path.getsize(edx)
    n_clumps = int(db.c.numRclumps)
    n_ent = C.c_uint64()
    capi._chk(capi.lib().bhip_acx_export(dev._h, None, None, None, 0, C.byref(n_ent), None, 0, None))
    acx_entries = int(n_ent.value)
    acx_bytes = 5 + 4 * (1 << (2 * args.K)) + 3 * acx_entries          # what the LARGE-format file holds (the SMALL format, 2.5 B per entry, below 2^20 clumps)
    log("[bench] rank %d: db %d refs / %d clumps (.edx %.2f GB, accelerator %.2f G entries; read %.1f s, device upload + accelerator build %.1f s), %d reads -> %d unique (ingest %.1f s) on %s"
        % (rank, db.c.totR, db.c.numRclumps, edx_bytes / 1e9, acx_entries / 1e9, t_db, t_dev, qs.n_reads, qs.n_uniq, t_q, info["name"]))

    # pool batch b = unique queries [b U / P, (b+1) U / P).  The job = `steps` pool batches in a row (strong scaling: the same job
    # whatever N is); rank r aligns the r-th N-th of that sequence of unique queries -- every rank the same number of reads, in pieces
    # of at most one pool batch per device call, so the per-batch fixed costs are not multiplied by N
    U, P = qs.n_uniq, args.pool
    def pool_range(b):
        return (b * U // P, (b + 1) * U // P)
    weak = args.scaling == "weak"
    def job_share(first, count, r=rank, weak=weak):
        # strong scaling (default: the fixed job north_star asks about): the single-GPU job of `count` pool batches cut into N equal
        # shares of unique queries.  weak scaling: rank r aligns `count` pool batches of its own, starting r batches further into the
        # pool so that the ranks are not on the same reads at the same time
        if weak:
            return [pool_range((first + k + r) % P) for k in range(count)]
        return share_of_job([pool_range((first + k) % P) for k in range(count)], r, world)
    def job_reads(first, count, weak=weak):
        if weak and world > 1:
            return sum(reads_per_pool_batch[(first + k + r) % P] for r in range(world) for k in range(count))
        return sum(reads_per_pool_batch[(first + k) % P] for k in range(count))
    batch_uniq = max(1, U // P + 1)         # at most one pool batch per device call
    reads_per_pool_batch = [qs.reads_in(b * U // P, (b + 1) * U // P) for b in range(P)]

    # N > 1: the product's own exchange -- the library's RCCL communicator over the ranks (one process per GPU: the 128-byte id
    # travels through torch.distributed, which is plumbing here) and bh_search_multi, the function `burst_hip --gpus N` runs
    comm = None
    node = None
    def make_comm():
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            capi._chk(capi.lib().bhip_comm_unique_id(buf))
            idt = torch.tensor(list(buf), dtype=torch.uint8)
        idt = idt.cuda()
        dist.broadcast(idt, 0)
        idb = (C.c_uint8 * 128)(*idt.cpu().tolist())
        c = C.c_void_p()
        capi._chk(capi.lib().bhip_comm_create_rank(world, rank, local_rank, idb, C.byref(c)))
        return c
    rs = None
    def search(ranges, engine=None):
        """one job share through the product's scheduler; N > 1: + the hand-over of the records to rank 0"""
        if use_dist:
            return (engine or rs).search(qs, ranges, args.mode, batch_uniq)
        return host.align_ranges(dev, qs, ranges, args.mode, batch_uniq, run=_own)
    def timed(ranges, engine=None):
        """(seconds: the slowest rank's, barrier to barrier; records rank 0 holds; the Run)"""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t_ = time.time()
        run_ = search(ranges, engine)
        n_ = int(run_.c.nHits)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        e_ = time.time() - t_
        if use_dist:
            tt_ = torch.tensor([e_], dtype=torch.float64, device=pdev)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            e_ = float(tt_.item())
            nr_ = torch.tensor([n_ if rank == 0 else 0], dtype=torch.int64, device=pdev)
            dist.all_reduce(nr_)
            n_ = int(nr_.item())
        return e_, n_, run_

    # warm-up: sizes the library's grow-only buffers for this workload and runs W untimed steps
    # (the page-locked record buffer is allocated once, outside the timed region: the command line does it once per job as well)
    _own = host.Run()
    ent_per_step = min(batch_uniq, max((b - a for a, b in job_share(0, P)), default=1)) * (2 if args.fr else 1)
    share = max(1, max(sum(b - a for a, b in job_share(args.warmup, args.steps, weak=w_)) for w_ in ((False, True) if world > 1 else (weak,)))) * (2 if args.fr else 1)
    if use_dist:      # (the extra jobs of the N > 1 line use the same record buffers: configs[3]'s 10 M reads over the ranks)
        share = max(share, sum(b - a for a, b in job_share(args.warmup, max(1, int(round(10e6 / max(1, args.reads)))), weak=False)) * (2 if args.fr else 1))
    cap_rec = int(max(share, 4 * ent_per_step) * (4.0 if args.mode in ("FORAGE", "ALLPATHS") else 1.5)) + (1 << 20)
    if use_dist:
        if args.gather == "shm":
            # the ranks' record buffers are shared-memory segments rank 0 maps (bh_node.c): no collective on the data path.  The job
            # name travels through the launcher (plumbing); rank 0 opens first
            jt = torch.tensor([int.from_bytes(os.urandom(6), "little") if rank == 0 else 0], dtype=torch.int64, device=pdev)
            dist.broadcast(jt, 0)
            job = "bench%x" % int(jt.item())
            why = ""
            def try_open():
                try:
                    return host.Node(job, rank, world, cap_rec), ""
                except host.HostError as e:      # /dev/shm too small for the segment, or none
                    return None, str(e)
            if rank == 0:
                node, why = try_open()
            ok = torch.tensor([1 if (rank != 0 or node is not None) else 0], dtype=torch.int64, device=pdev)
            dist.broadcast(ok, 0)                # (rank 0 first: the others map its segment when they open theirs)
            if int(ok.item()) and rank != 0:
                node, why = try_open()
            ok = torch.tensor([1 if node is not None else 0], dtype=torch.int64, device=pdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):               # some rank has no segment: every rank takes the RCCL gather instead
                if node is not None:
                    node.close()
                    node = None
                if one_dev is not None:
                    raise SystemExit("BURST_BENCH_DEVICE: the shared-memory hand-over is the only one for ranks on one device (%s)" % why)
                log("[bench] rank %d: no shared-memory hand-over%s: RCCL gather" % (rank, " (%s)" % why if why else ""))
                args.gather = "rccl"
        if args.gather == "rccl":
            comm = make_comm()
        rs = host.RankSearch(dev, rank, world, comm, node=node)
        rs.reserve(cap_rec)
        if rank == 0 and args.gather == "rccl":      # the buffer the gathered records land in: made here, as burst_hip does in its "batch buffers" phase
            rs.reserve_all(world * cap_rec, pinned=True)      # (shared memory: rank 0 reads the ranks' segments where they lie, nothing to reserve)
    else:
        _own.reserve(cap_rec)
    # (bhip_reserve = the command line's "batch buffers" phase: device buffers for this batch size + the library's own warm-up pass)
    if not args.no_prime:
        dev.reserve(int(ent_per_step), int(args.read_len))
    # one priming call of four batches (setup, not a warm-up step): the first call long enough to have two batches' staging copies
    # queued when a batch's records are handed over pays ~17 ms once per process inside the runtime's asynchronous copy (seen with
    # --warmup 1 in front of the timed region's second batch)
    if not args.no_prime:
        search([pool_range(k % P) for k in range(4)] if world == 1 else job_share(0, 4 if weak else 4 * world))
    w_run = search(job_share(0, max(1, args.warmup)))
    # The record buffer is sized from what the warm-up found: a database with strain-level redundancy returns twenty equally good references
    # per read where the default one returns one, and a buffer that has to grow inside the timed region (page-locked memory, a copy of
    # everything held so far) would be the measurement.  The command line sizes its buffer once per job as well.
    if not use_dist:
        per_step = int(w_run.c.nHits) / max(1, args.warmup)
        need_rec = int(per_step * args.steps * 1.15) + (1 << 20)
        if need_rec > cap_rec:
            cap_rec = need_rec
            _own.reserve(cap_rec)
            search(job_share(0, 1))          # (one batch into the new buffer: its pages touched outside the timed region)
    elapsed, n_records, run = timed(job_share(args.warmup, args.steps))
    total_reads = job_reads(args.warmup, args.steps)
    own_main = rs.own_stats() if use_dist else (run.stats(), int(run.c.nBatches), float(run.c.secAlign))          # (this rank's counters of the timed job: the jobs below overwrite them)
    handover_main = None
    if use_dist and rank == 0:      # what rank 0 holds after the timed search, read once through (outside the timed region): every rank's run, its records
        runs_ = rs.view.runs()
        handover_main = {"kind": "view over the ranks' shared-memory segments" if (node is not None and rs.view.n_runs == world and world > 1) else "one array",
                         "records_per_run": [int(len(x)) for x in runs_], "distinct_entries": int(sum(len(np.unique(x["q"])) for x in runs_)),
                         "xor_of_reference_numbers": int(np.bitwise_xor.reduce(np.concatenate([x["refIx"] for x in runs_]))) if sum(len(x) for x in runs_) else 0}
    # what the records are for: rank 0's consolidation of the timed job's records into .b6 lines (bh_report_view: per-mode selection,
    # coordinates, formatting), written to /dev/null -- outside the timed region, reported beside it
    consolidation = None
    # (the timed job cycles through the pool: a query appears several times in its records, which no report accepts; one pass over the pool)
    cons_run = search(job_share(0, P, weak=False))
    if rank == 0:
        try:
            t_ = time.time()
            n_lines_ = host.report_view(os.devnull, db, qs, rs.view, args.mode, 0) if use_dist else host.report(os.devnull, db, qs, cons_run.hits, args.mode, 0)
            consolidation = {"seconds": time.time() - t_, "lines": int(n_lines_), "reads": int(qs.n_reads),
                             "what": "rank 0: bh_report over the records of ONE pass over the read pool (every unique query once; all ranks' records), .b6 lines to /dev/null; not part of `value`"}
        except Exception as e:
            consolidation = {"error": str(e)}
    # N > 1: the same ranks again on (a) the other scaling, (b) BASELINE configs[3]'s job -- 10 M reads over all GPUs, one or two batches
    # per rank --, (c) the main job with the records gathered over RCCL (bhip_comm_gather_hits across all N ranks) instead of
    # meeting in shared memory.  Extra keys of the line; `value` stays the main job's
    extra = {}
    if use_dist:
        if world > 1:
            e2, n2, _ = timed(job_share(args.warmup, args.steps, weak=not weak))
            r2 = job_reads(args.warmup, args.steps, weak=not weak)
            extra["weak_scaling" if not weak else "strong_scaling"] = {"value": r2 / e2, "unit": "reads/s", "reads": r2, "seconds": e2, "records": n2,
                "what": ("every rank aligns %d batches of its own: the job is N times the single-GPU job" % args.steps) if not weak else "the single-GPU job cut into N equal shares"}
        nb3 = max(1, int(round(10e6 / max(1, args.reads))))
        search(job_share(0, nb3, weak=False))
        e3, n3, _ = timed(job_share(args.warmup, nb3, weak=False))
        r3 = job_reads(args.warmup, nb3, weak=False)
        extra["configs3_job"] = {"value": r3 / e3, "unit": "reads/s", "reads": r3, "seconds": e3, "records": n3, "reads_per_rank": r3 // world,
                                 "what": "BASELINE configs[3]'s job size: %d reads cut over %d GPUs (strong scaling)" % (r3, world)}
    # The RCCL gather across all ranks in the same run (second measurement of the main job, records through bhip_comm_gather_hits inside the
    # timed region).  It runs LAST -- rank 0 has the whole line ready by then -- and under a watchdog: this exchange has never met two
    # devices (every box so far had one), and a collective that does not come back must not take the run's measurement with it.
    want_rccl = use_dist and one_dev is None and args.gather == "shm"

    def rccl_extra():
        comm2 = make_comm()
        rs2 = host.RankSearch(dev, rank, world, comm2, node=None)
        rs2.reserve(cap_rec)
        if rank == 0:
            rs2.reserve_all(world * cap_rec, pinned=True)
        search(job_share(0, max(1, args.warmup)), rs2)
        e4, n4, _ = timed(job_share(args.warmup, args.steps), rs2)
        al = torch.tensor([float(rs2.mr.secSearch)], dtype=torch.float64, device=pdev)
        dist.all_reduce(al, op=dist.ReduceOp.MAX)
        gp = int(rs2.mr.gatherPath)      # which gather this rank's records really took (BhMultiRank.gatherPath, set by bh_search_multi_ex)
        out = {"rccl_ranks": world, "value": total_reads / e4, "unit": "reads/s", "seconds": e4, "records": n4, "rccl_gather_ms": max(0.0, e4 - float(al.item())) * 1e3,
               "records_from": ("device (every batch's records staged into the send buffer while resident: bhip_comm_stage_device -> bhip_comm_gather_staged)" if gp == 1 else
                                "host (the staged gather was refused, the host copy went up again: bhip_comm_gather_hits)" if gp == 2 else "no collective ran (gatherPath 0)"),
               "gather_path": gp,
               "what": "the main job with the records gathered to rank 0 by bhip_comm_gather_staged (ncclAllGather of the counts + grouped ncclSend/ncclRecv over xGMI from the ranks' device-resident "
                       "records, one copy to rank 0's host) inside the timed region instead of the shared-memory hand-over; rccl_gather_ms = that time minus the slowest rank's align phase"}
        rs2.close()
        capi.lib().bhip_comm_destroy(comm2)
        return out

    def rccl_guarded(line_so_far):
        """rccl_extra under a watchdog; when it does not answer in time rank 0 prints the line it has (line_so_far) and every rank leaves"""
        import threading
        limit = float(os.environ.get("BURST_BENCH_RCCL_TIMEOUT", "240")) + (0.0 if rank == 0 else 120.0)      # (the other ranks wait for rank 0 first)

        def give_up():
            if rank == 0 and line_so_far is not None:
                line_so_far["rccl"] = {"error": "the RCCL gather across %d ranks did not answer within %.0f s (the line's other figures were measured before it)" % (world, limit)}
                print(json.dumps(line_so_far), flush=True)
            os._exit(0)
        t = threading.Timer(limit, give_up)
        t.daemon = True
        t.start()
        try:
            return rccl_extra()
        except Exception as e:
            return {"error": str(e)}
        finally:
            t.cancel()
    rccl_early = None
    if want_rccl and world == 1:
        rccl_early = rccl_guarded(None)          # (one process under torch.distributed.run, BURST_BENCH_DIST1: the device is closed further down)
    elif want_rccl and rank != 0:
        rccl_guarded(None)          # (waits in the communicator's creation until rank 0 has put its line together)

    if rank == 0 and os.environ.get("BHIP_PROF"):      # library built with EXTRA_HIPFLAGS=-DPFM_PROF: wave-cycles per prefilter phase
        import ctypes
        lib = capi.lib()
        if hasattr(lib, "bhip_debug_prof"):
            arr = (ctypes.c_ulonglong * 8)()
            lib.bhip_debug_prof(arr, 1)
            tot = float(sum(arr)) or 1.0
            log("[bench] prefilter phase share: " + " ".join("%d:%.1f%%" % (i, 100.0 * v / tot) for i, v in enumerate(arr)) + "  total wave-cycles %.3g" % tot)
    if rank == 0:
        st, nb, sec_align = own_main
        nb = max(1, nb)
        per = lambda k: float(st[k]) / nb
        two_stage = st["prefix_words"] > 0
        masked = st["prefilter_launches"] > 0
        # Algorithmic bytes per kernel launch (SURVEY.md 8d figures): prefilter = 8 B offset pair per
        # sampled word + 3 B per list entry (the .acx size) + 8 B per emitted task; column sweeps = 0.5 B (one 4-bit
        # symbol) per swept column of one reference lane + len/2 B of query and 12 B of result per unit.
        kernels = {}
        if masked and st["ms_prefilter_hash"] > 0:
            n = max(1, st["prefilter_launches"])
            pf_name = {0: "k_prefilter_cf", 2: "k_prefilter_cw", 3: "k_prefilter_cq"}.get(st["prefilter_algo"], "k_prefilter_mask")
            kernels[pf_name] = (st["ms_prefilter_hash"] / n, (8.0 * st["n_seed_words"] + 3.0 * st["acx_entries_read"] + 8.0 * st["n_lane_tasks"]) / n, "hbm")
            kernels["k_seed_ranges"] = (st["ms_seed"] / n, (8.0 * st["n_seed_words"] + 8.0 * st["n_seed_words"]) / n, "hbm")
        n = max(1, st["myers_launches"])
        if two_stage:
            cols = st["n_task_columns"] if masked else st["n_columns"] * 16
            units = st["n_lane_tasks"] if masked else st["n_pairs"] * 16
            kernels["k_myers_prefix%s<%d>" % ("_task" if masked else "", st["prefix_words"])] = (st["ms_myers_prefix"] / n, (0.5 * cols + units * (args.read_len / 2.0 + 12.0)) / n, "valu")
            nw_ = (args.read_len + 31) // 32      # three words and more: the banded kernel takes the windows (k_myers_window_band<2> at E <= 9)
            kernels["k_myers_window_band<2>" if nw_ >= 3 else "k_myers_window<%d>" % nw_] = (st["ms_myers_window"] / n, (0.5 * st["n_window_columns"] + st["n_windows"] * (args.read_len / 2.0 + 12.0)) / n, "valu")
        else:
            kernels["k_myers<%d>" % ((args.read_len + 31) // 32)] = (st["ms_myers"] / n, st["bytes_algorithmic"] / n, "valu")
        kernels["k_rescore_*"] = (st["ms_rescore"] / nb, (st["n_raw_hits"] * (20.0 + args.read_len / 2.0 + (args.read_len + 16) / 2.0) + st["n_hits"] * 20.0) / nb, "hbm")
        tot_ms = lambda k: kernels[k][0] * (st["prefilter_launches"] if k.startswith("k_prefilter") or k == "k_seed_ranges" else nb if k == "k_rescore_*" else n)
        # dominant = the longest kernel ON THE CRITICAL PATH (prefilter -> sweeps -> re-scoring): k_seed_ranges works for the NEXT batch, on its
        # own stream and deliberately with a few blocks per CU (option seed_ahead_blocks), so its elapsed time says how slowly it was allowed
        # to run beside the chain, not what bounds the step; it stays in per_kernel, marked off_critical_path
        dom = max((k for k in kernels if k != "k_seed_ranges"), key=tot_ms)
        ms_dom, bytes_dom, bound_dom = kernels[dom]
        achieved = bytes_dom / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        pmc, pmc_source = pmc_table([k for k in kernels if k != "k_seed_ranges"])
        try:
            mix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "valu_mix.json")))
        except Exception:
            mix = {}
        def valu_weight(kname):
            base = kname.split("<")[0].replace("_*", "_reg")
            for mk, mv in mix.items():
                if mk.split("<")[0] == base:
                    return mv.get("weight")
            return None
        def pmc_of(kname):
            return pmc.get(kname.split("<")[0].replace("_*", "_reg"), {})
        per_kernel = {}
        for k, v in kernels.items():
            e = {"ms_per_launch": v[0], "algorithmic_bytes_per_launch": v[1], "GBps": (v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0), "bound": v[2]}
            pk = pmc_of(k)
            if pk.get("hbm_bytes_per_launch") is not None:
                e["traffic"] = pk["hbm_bytes_per_launch"]
                if v[2] == "hbm" and pk.get("hbm_bytes_per_launch_gather_calibrated"):      # the other bound: one 64-byte half per requested line (TRAFFIC_CALIBRATION below)
                    e["traffic_half_lines"] = pk["hbm_bytes_per_launch_gather_calibrated"]
            if pk.get("valu_frac") is not None:
                e["valu_frac"] = pk["valu_frac"]
                w = valu_weight(k)
                if w:      # tools/valu_mix.py: half-rate VOP3 forms cost two issue slots, and the clock under load is 2.05 GHz
                    e["valu_issue_frac"] = pk["valu_frac"] * w
            if k == "k_seed_ranges":
                e["off_critical_path"] = True
            per_kernel[k] = e
        cells = (st["n_task_columns"] if masked else st["n_columns"] * 16.0) * min(args.read_len, 32.0 * max(1, st["prefix_words"])) + st["n_window_columns"] * float(args.read_len)
        ms_sweeps = st["ms_myers"]
        scale_to_metric = 31.5e9 / max(1, edx_bytes)
        rec_per_read = st["acx_entries_read"] / max(1.0, float(st["n_queries"]))
        # the metric's database is 31.5 GB of .edx: how this run's database compares, and where the bench line measured AT that size is kept
        # (profiles/r04_sizes.json: three sizes, same kernels; round 3 could only extrapolate -- 5-byte records did not fit the device)
        extrap = {"metric_database": "31.5 GB RefSeq .edx", "this_edx_bytes": edx_bytes, "size_ratio": scale_to_metric,
                  "acx_entries_here": acx_entries, "acx_records_per_read_here": rec_per_read, "this_run_is_at_metric_size": edx_bytes >= 31.0e9}
        try:
            extrap["measured_sizes"] = json.load(open(os.path.join(ROOT, "profiles", "r06_sizes.json")))
        except Exception:
            extrap["measured_sizes"] = None
        res = {
            "metric": "aligned reads/sec (node), %d-bp synthetic reads @%s id vs RefSeq stand-in .edx/.acx (DB%d), -m %s; %d GPU" % (args.read_len, args.id, args.K, args.mode, world),
            "value": total_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32 bit-vectors (u8 edit distances)", "data": "synthetic",
            "config": {"workload": "%s on %d GPU(s): %d synthetic %d-bp reads per step (%s edits%s%s), -m %s -i %s, vs %d references x %d bp "
                                   "(%.2f Gbp; %s%d clumps; .edx %.2f GB + DB%d .acx %.2f GB); every step stages its batch afresh through the product's batch scheduler"
                                   % (shape_name(args), world, args.reads, args.read_len, args.edits.replace(",", "/"), ", both strands" if args.fr else "", (", %g of the bases IUPAC codes" % args.iupac) if args.iupac else "",
                                      args.mode, args.id, args.n_base * args.n_variants, args.ref_len,
                                      args.n_base * args.n_variants * args.ref_len / 1e9,
                                      "" if args.db_profile == "pairs" else "--db-profile strains: 70 % pairs at 5 %, 30 % families of 60 / 200 / 500 strains at 1 / 0.5 / 0.1 %; ",
                                      db.c.numRclumps, edx_bytes / 1e9, args.K, acx_bytes / 1e9),
                       "db_profile": args.db_profile,
                       "parallelism": ("query-sharded x%d (%s, in device batches of up to %d), DB replicated; " % (world, "weak scaling: every rank aligns %d batches of its own, the job is N times the single-GPU job" % args.steps
                                                                                                                     if weak else "strong scaling: rank r aligns the r-th N-th of the single-GPU job's unique queries", batch_uniq)) +
                                      ("no collective on the data path: every rank's record buffer is a page-locked shared-memory segment (its batches' records land there over its own PCIe link, behind the "
                                       "batch), rank 0 has the segments mapped side by side and reads the records where they lie (bh_node.c inside bh_search_multi_ex; bh_report_view consumes such a view) -- no copy" if use_dist and args.gather == "shm" else
                                       "one RCCL gather of the hit records to rank 0 (bhip_comm_gather_hits inside bh_search_multi, the function burst_hip --gpus N --gather rccl runs)"),
                       "timed_region": "bh_align_ranges over %d batches (copies + device routing two batches ahead, seed lookups + profiles one batch ahead, alignment, records to host memory)%s" %
                                       (nb, (" + hand-over of all ranks' records to rank 0 (shared memory)" if args.gather == "shm" else " + RCCL gather") if use_dist else ""),
                       "extrapolation": extrap,
                       "device": info["name"], "n_cu": info["n_cu"],
                       # which exchange built the accelerator (round 5's verdict: "did RCCL see N ranks" must be answerable for both collectives): the
                       # bench's ranks each build the whole accelerator on their own device (outside the timed region: host.device_upload_s) -- no exchange;
                       # the cooperative build (bhip_build_accelerator_shared over bhip_comm_share / RCCL broadcasts) is what burst_hip --gpus N and
                       # python -m burst_amd.run use, measured on one device only
                       "accelerator_build": {"by": "every rank on its own device (bhip_init with K)" if not args.acx_file else "loaded from the .acx file", "exchange": None, "ranks_in_exchange": 0}},
            "roofline": {"bound": "hbm" if bound_dom == "hbm" else "valu", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": pmc_of(dom).get("hbm_bytes_per_launch"), "pmc_source": pmc_source,
                         "traffic_half_lines": pmc_of(dom).get("hbm_bytes_per_launch_gather_calibrated") if bound_dom == "hbm" else None,
                         "traffic_calibration": dict(TRAFFIC_CALIBRATION, device_record_bytes_per_launch=(
                             (st["acx_entries_read"] * 4.0 + st["n_seed_words"] * 8.0 + st["n_lane_tasks"] * 8.0) / max(1, st["prefilter_launches"]) if dom.startswith("k_prefilter") else None)),
                         "note": "achieved / frac are measured by THIS run (HIP events); traffic, valu_frac and the other counter-derived fields come from the kept profiling pass named in pmc_source (an earlier run of the same kernels), null when that table is refused. Dominant kernel by time on the critical path (HIP events on its stream; k_seed_ranges works for the next batch beside the chain, throttled: off_critical_path in per_kernel); per_kernel gives each kernel's own bound: the prefilter and the re-scorer are "
                                 "bound by HBM/LDS latency of short gathers, the k_myers_* sweeps by integer VALU issue (valu_frac = issued VALU instructions x 2 cycles / peak, from the PMC pass; half-rate VOP3 forms count once). traffic follows the guide's 2 x FETCH_SIZE rule = the bytes of every 128-byte line the kernel REQUESTED (an upper bound of what moved); traffic_half_lines = FETCH_SIZE + WRITE_SIZE = one 64-byte half per requested line (the lower bound): see traffic_calibration (round 6; rounds 2-5 read the second as 'what a sector gather really moves', profiles/r02k_fetch_calibration.txt)",
                         "algorithmic_bytes_per_launch": bytes_dom, "ms_per_launch": ms_dom,
                         "per_kernel": per_kernel,
                         "gcups_sweeps": cells / (ms_sweeps * 1e-3) / 1e9 if ms_sweeps > 0 else 0.0},
            "phases_ms_per_batch": {k: per(k) for k in ("ms_h2d", "ms_stage_copy", "ms_stage_route", "ms_peq", "ms_prefilter", "ms_seed", "ms_prefilter_hash", "ms_myers", "ms_myers_prefix", "ms_myers_window", "ms_rescore", "ms_d2h", "ms_total")},
            "work": {"records": n_records, "entries_per_batch": st["n_queries"] / nb, "raw_hits": st["n_raw_hits"], "hits": st["n_hits"],
                     "acx_entries_per_read": st["acx_entries_read"] / max(1.0, float(st["n_queries"])), "windows": st["n_windows"], "window_columns": st["n_window_columns"],
                     "lane_tasks_per_read": st["n_lane_tasks"] / max(1.0, float(st["n_queries"])), "task_columns": st["n_task_columns"],
                     "raw_hits_per_read": st["n_raw_hits"] / max(1.0, float(st["n_queries"])), "records_per_read": st["n_hits"] / max(1.0, float(st["n_queries"])),
                     "seed_words_per_read": st["n_seed_words"] / max(1.0, float(st["n_queries"]))},
            "host": {"db_read_s": t_db, "device_upload_s": t_dev, "query_ingest_s": t_q, "sec_in_device_calls": sec_align},
        }
        if world == 1 and not args.no_short_job:
            # One rank's share of BASELINE configs[3]'s job on 8 GPUs: 10 M reads / 8 = 1.25 M reads = less than one batch.  A job that
            # short has nothing to hide its staging, seed lookups and match profiles behind.  Whole, and cut into four pieces so that
            # piece k + 1 is prepared while piece k is aligned (bh_align.c, BURST_HOST_PIECES): the pieces' fixed costs eat the overlap.
            try:
                u_short = max(1, int(U * min(1.0, 1250000.0 / max(1, qs.n_reads))))
                sj = {"reads": qs.reads_in(0, u_short), "what": "one job of 1.25 M reads (the share of one of 8 GPUs of configs[3]'s 10 M reads) through bh_align_ranges: "
                      "wall ms from the call to the last record in host memory; whole = one batch (the scheduler's default), in_pieces = cut into four (BURST_HOST_PIECES=4), which does not pay"}
                for label, pieces in (("ms_whole", "1"), ("ms_in_pieces", "4")):
                    os.environ["BURST_HOST_PIECES"] = pieces
                    search([(0, u_short)])
                    best_ = None
                    for _ in range(5):
                        torch.cuda.synchronize(); t_ = time.time(); search([(0, u_short)]); torch.cuda.synchronize()
                        e_ = (time.time() - t_) * 1e3
                        best_ = e_ if best_ is None else min(best_, e_)
                    sj[label] = best_
                    if pieces == "1":      # where the one-batch job's device time goes (HIP events of one more run): nothing overlaps here, so the phases add up
                        st_ = search([(0, u_short)]).stats()
                        sj["phases_ms_whole"] = {k: float(st_[k]) for k in ("ms_h2d", "ms_stage_copy", "ms_stage_route", "ms_seed", "ms_peq", "ms_prefilter_hash", "ms_myers_prefix", "ms_myers_window", "ms_rescore", "ms_d2h", "ms_total")}
                os.environ.pop("BURST_HOST_PIECES", None)
                res["one_rank_share_of_configs3"] = sj
                log("[bench] 1.25 M-read job: %.2f ms whole, %.2f ms in four pieces" % (sj["ms_whole"], sj["ms_in_pieces"]))
            except Exception as e:
                res["one_rank_share_of_configs3"] = {"error": str(e)}
        if world == 1 and args.ab_host:
            # the same job with the scheduler staging the batch after next BETWEEN two batches (device idle) instead of from the
            # library's "chain enqueued" hook (device busy): BURST_HOST_NO_HOOK
            res["ab_host"] = []
            for label, env in (("no_enqueued_hook", {"BURST_HOST_NO_HOOK": "1"}), ("default", {})):
                for k_, v_ in env.items():
                    os.environ[k_] = v_
                search(job_share(0, max(1, args.warmup)))
                torch.cuda.synchronize(); ta = time.time(); search(job_share(args.warmup, args.steps)); torch.cuda.synchronize()
                ea = time.time() - ta
                res["ab_host"].append({"variant": label, "value": total_reads / ea, "ms_per_step": ea / args.steps * 1e3})
                log("[bench] ab_host %s: %.1f M reads/s, %.3f ms per step" % (label, total_reads / ea / 1e6, ea / args.steps * 1e3))
                for k_ in env:
                    os.environ.pop(k_, None)
        if world == 1 and args.ab:
            # A/B on the resident database: the same warm-up and steps under other tuning options, the default options' line again at the end
            defaults = {"prefilter_cw": 2, "prefilter_bytes": 1, "prefilter_rb": 0, "seed_min_need": -1, "seed_drop_len": 8, "prefilter_table": 0, "prefilter_waves": 0, "prefilter_algo": -1, "prune": 1, "oversub": 2, "band": 1, "seed_ahead": 1, "seed_ahead_blocks": 2, "peq_ahead_blocks": 16, "sweep_blocks": 8, "lanes": 1}
            res["ab"] = []
            for spec in list(args.ab) + [""]:
                kv = dict(x.split("=") for x in spec.split(",") if x)
                for name, val in kv.items():
                    dev.set_option(name, int(val))
                search(job_share(0, max(1, args.warmup)))
                torch.cuda.synchronize()
                ta = time.time()
                r2 = search(job_share(args.warmup, args.steps))
                torch.cuda.synchronize()
                ea = time.time() - ta
                st2, nb2 = r2.stats(), max(1, int(r2.c.nBatches))
                res["ab"].append({"opts": spec or "(defaults again)", "value": total_reads / ea, "ms_per_step": ea / args.steps * 1e3, "records": int(r2.c.nHits),
                                  "acx_entries_per_read": st2["acx_entries_read"] / max(1.0, float(st2["n_queries"])), "lane_tasks_per_read": st2["n_lane_tasks"] / max(1.0, float(st2["n_queries"])),
                                  "ms_prefilter_kernel": st2["ms_prefilter_hash"] / max(1, st2["prefilter_launches"]), "ms_myers": st2["ms_myers"] / nb2, "ms_rescore": st2["ms_rescore"] / nb2,
                                  "prefilter_algo": st2["prefilter_algo"]})
                log("[bench] ab %s: %.1f M reads/s, %.3f ms per step, prefilter kernel %.3f ms, %.1f records per read, %.2f lane tasks per read, %d records"
                    % (spec or "(defaults)", total_reads / ea / 1e6, ea / args.steps * 1e3, res["ab"][-1]["ms_prefilter_kernel"], res["ab"][-1]["acx_entries_per_read"], res["ab"][-1]["lane_tasks_per_read"], int(r2.c.nHits)))
                for name in kv:
                    if name in defaults:
                        dev.set_option(name, defaults[name])
        res.update(extra)
        res["consolidation"] = consolidation
        res["cpu_baseline"] = None
        ref_note = None
        if want_base:      # N = 1 only (the contract); the other ranks would sit in the barrier meanwhile
            import shutil
            ram_backed = args.workdir.startswith("/dev/shm")
            # (the sizes of THIS database, not the estimate the size was picked with: other read lengths shear the references differently)
            need_now = (edx_bytes if ram_backed else 0) + reference_memory(edx_bytes, acx_bytes, args.K) + 16e9
            fits = need_now <= memory_limit() * 0.92
            if not fits:
                res["cpu_baseline_skipped"] = ("the reference's accelerated run needs %.0f GB of host memory at this database size (.edx %.1f GB + .acx %.1f GB in its memory%s); "
                                               "this job may use %.0f GB" % (need_now / 1e9, edx_bytes / 1e9, acx_bytes / 1e9,
                                                                             ", the .edx file in a RAM-backed directory" if ram_backed else "", memory_limit() / 1e9))
                log("[bench] " + res["cpu_baseline_skipped"])
                try:
                    db.close()
                    ref_note = exhaustive_reference_sample(edx, reads_fa, args, n_clumps)
                    db = host.Db.read(edx, None, K=args.K)
                except Exception as e:
                    ref_note = {"error": str(e)}
            else:
                # the reference reads an .acx FILE: a named pipe, fed from the tables the device built while the reference reads it
                try:
                    t = time.time()
                    db.close()          # (this process's copy of the database: the reference holds its own, and the memory is shared)
                    room = memory_limit() - memory_in_use()
                    if memory_in_use() and room < reference_memory(edx_bytes, acx_bytes, args.K) + (8 << 30):
                        res["cpu_baseline_skipped"] = "%.0f GB of the job's memory are free, the reference needs %.0f" % (room / 1e9, reference_memory(edx_bytes, acx_bytes, args.K) / 1e9 + 8)
                    else:
                        res["cpu_baseline"] = cpu_baseline(edx, acx + ".pipe", reads_fa, args, AcxFeed(dev, n_clumps, args.K, acx + ".pipe"))
                        if res["cpu_baseline"] is None and getattr(cpu_baseline, "too_small", None):
                            res["cpu_baseline_skipped"] = cpu_baseline.too_small
                            ref_note = {"what": "the reference WITH its accelerator, same .edx/.acx (its timing was not resolvable: cpu_baseline_skipped)"}
                        if res["cpu_baseline"]:
                            res["cpu_baseline"]["acx"] = ("%.1f GB .acx streamed to the reference through a named pipe from the device-built tables (bh_acx_write_from_device, %.0f s of its %.0f s run): "
                                                          "it never exists as a file" % (acx_bytes / 1e9, getattr(cpu_baseline, "feed_s", 0.0) or 0.0, getattr(cpu_baseline, "wall", 0.0)))
                    db = host.Db.read(edx, None, K=args.K)
                    log("[bench] reference on the host cores: %.1f s (peak memory of the job so far: %s)" % (time.time() - t, memory_peak_str()))
                except Exception as e:
                    res["cpu_baseline_skipped"] = "reference run failed: %s" % e
        if res["cpu_baseline"]:
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        if res["cpu_baseline"] or (ref_note and "error" not in ref_note):
            try:
                res["parity_vs_reference"] = parity_vs_reference(dev, db, args, batch_uniq)
                if ref_note and res["parity_vs_reference"]:
                    res["parity_vs_reference"]["reference_run"] = ref_note
                    res["parity_vs_reference"]["what"] = res["parity_vs_reference"]["what"].replace("oracle/_ref/burst%d" % args.K, "oracle/_ref/burst%d without -a (exhaustive)" % args.K)
            except Exception as e:      # reported, never fatal for the measurement
                res["parity_vs_reference"] = {"error": str(e)}
        elif ref_note:
            res["parity_vs_reference"] = ref_note
        if handover_main is not None:
            res["handover"] = handover_main
        if world == 1 and (not args.no_end_to_end or (not args.no_continuity and args.db_scale > 1.5)):
            # the command line and the small-database run are processes of their own with a database of their own on the device: this
            # process lets go of its copy first (a large database does not fit twice)
            if rs is not None:
                rs.close(); rs = None
            _own.close(); _own = None
            dev.close()
            db.close()
            try:      # (what the processes that follow find on the device: this one should have given everything back)
                torch.cuda.synchronize()
                f_, t_ = torch.cuda.mem_get_info(local_rank)
                res["device_memory_after_release"] = {"free_GB": f_ / 1e9, "total_GB": t_ / 1e9}
                log("[bench] device memory after this process let go of its database: %.1f GB free of %.1f" % (f_ / 1e9, t_ / 1e9))
            except Exception:
                pass
        if world == 1 and not args.no_end_to_end:
            try:
                res["end_to_end"] = end_to_end(edx, reads_fa, args, local_rank)
            except Exception as e:      # reported, never fatal for the measurement
                res["end_to_end"] = {"error": str(e)}
            if res["end_to_end"] and "reads_per_s" in res["end_to_end"]:
                res["end_to_end_reads_per_s"] = res["end_to_end"]["reads_per_s"]
        if world == 1 and not args.no_continuity and args.db_scale > 1.5:
            # rounds 1-3 quoted the rate on the 2.77 GB database: the same steps on it, for continuity (a child process; extra key)
            try:
                t = time.time()
                cmd = [sys.executable, os.path.abspath(__file__), "--db-scale", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--reads", str(args.reads), "--pool", str(args.pool),
                       "--no-cpu-baseline", "--no-end-to-end", "--no-continuity", "--workdir", os.path.join(args.workdir, "small")] + [x for kv in args.opt for x in ("--opt", kv)]
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                res["continuity_small_db"] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
                                              "acx_entries_per_read": d["work"]["acx_entries_per_read"], "seconds_spent": time.time() - t,
                                              "what": "the database of rounds 1-3 (BENCH_r03: 556 M reads/s), same steps, same kernels"}
            except Exception as e:
                res["continuity_small_db"] = {"error": str(e)}
        if world == 1 and not args.no_strains and args.db_profile == "pairs" and args.db_scale > 1.5:
            # The redundancy regime, driver-timed (round 5's verdict: the headline's stand-in is low-redundancy on purpose; a database with
            # families of near-identical strains moves the work from the prefilter to the sweeps and the rate by an order of magnitude, while
            # the reference does not move).  A child process with its own database; its reference leg and parity check included.
            try:
                t = time.time()
                cmd = [sys.executable, os.path.abspath(__file__), "--db-profile", "strains", "--db-scale", str(args.strains_scale), "--steps", str(max(4, args.steps // 2)), "--warmup", str(min(args.warmup, 3)),
                       "--reads", str(args.reads), "--pool", str(args.pool), "--cpu-sample", "100000", "--no-end-to-end", "--no-continuity", "--no-strains", "--no-short-job",
                       "--workdir", os.path.join(args.workdir, "strains")] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + [x for kv in args.opt for x in ("--opt", kv)]
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                cb = d.get("cpu_baseline") or {}
                res["strains"] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
                                  "lane_tasks_per_read": d["work"]["lane_tasks_per_read"], "lanes_within_budget_per_read": d["work"]["raw_hits_per_read"], "records_delivered_per_read": d["work"]["records_per_read"],
                                  "phases_ms_per_batch": d["phases_ms_per_batch"],
                                  "reference_reads_per_s": cb.get("value"), "reference_cores": cb.get("cores"), "parity_vs_reference": d.get("parity_vs_reference"),
                                  "seconds_spent": time.time() - t,
                                  "what": "--db-profile strains (70 % of the content as the headline's pairs, 30 % in families of 60 / 200 / 500 strains at 1 / 0.5 / 0.1 % divergence), same reads per step, "
                                          "-m BEST with the choice made on the device; the 30 GB run of this profile is kept under profiles/"}
                import shutil
                shutil.rmtree(os.path.join(args.workdir, "strains"), ignore_errors=True)
            except Exception as e:
                res["strains"] = {"error": str(e)}
        if want_rccl:
            res["rccl"] = rccl_early if world == 1 else rccl_guarded(res)
            # (the driver's parsed view keeps `config` and the top-level scalars: whether RCCL saw N ranks must be answerable from there)
            if isinstance(res["rccl"], dict):
                for k_ in ("rccl_ranks", "rccl_gather_ms"):
                    res[k_] = res["config"][k_] = res["rccl"].get(k_)
                res["config"]["rccl_records_from"] = res["rccl"].get("records_from")
        print(json.dumps(res), flush=True)
    if not args.keep_files and rank == 0 and args.db_scale >= 2:      # a RAM-backed work directory is given back
        import shutil
        shutil.rmtree(args.workdir, ignore_errors=True)
    if rs is not None:
        rs.close()
    if _own is not None:
        _own.close()
    if comm:
        capi.lib().bhip_comm_destroy(comm)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
