#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the BURST alignment hot path on MI355X, on the metric's own shape.

Workload (BASELINE.json configs[3] / `metric`, at the size one GPU box can build in a bench run): synthetic 100-bp reads
(0-2 edits, LLsim-style) against a RefSeq stand-in database with a DB15 (K = 15) accelerator, `-m BEST -i 0.98`.  The real
31.5 GB RefSeq subset is not reachable offline; the stand-in is a low-redundancy synthetic database (default 1.6 M random base
sequences x 2 variants at 5 % divergence = 3.2 M references, 4.5 Gbp: every 15-mer has ~4 unrelated list entries, as a
collision-dominated RefSeq-scale DB15 would have many more), its accelerator BUILT ON THE DEVICE from the .edx (no .acx is read
or uploaded), and `config.extrapolation` carries the slope measured over several database sizes (profiles/r03_slope.json).

A step = one batch of reads through the WHOLE device path as the product runs it: bench.py calls the C batch scheduler of
the `burst_hip` command line (bh_align_ranges, burst_amd/csrc/host/bh_align.c) -- each step's batch is staged afresh from
host memory (copies + device-side routing on the staging stream, one batch ahead of the batch being aligned), aligned
(seeds -> prefilter -> two-stage bit-parallel edit distance -> re-scoring -> sorted records) and its records handed back to
host memory behind the call.  Staging is therefore INSIDE the timed region.  The steps cycle through --pool distinct
batches of the sorted unique queries.  N > 1 (one process per GPU under torch.distributed.run): the database is replicated,
every rank aligns --steps batches of its own (weak scaling, the default: the job is N times the single-GPU job; --scaling strong
cuts the single-GPU job into N equal shares instead) through the product's multi-rank search (bh_search_multi_ex); the path partitions, so there is no collective on the data path -- every rank's page-locked record
buffer is a shared-memory segment rank 0 has mapped (bh_node.c), the hand-over inside the timed region is one word per rank and
rank 0 reads the records where they lie (`handover` in the JSON line).  --gather rccl times the RCCL gather instead;
torch.distributed only carries the job name and the barriers around the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel by time on the critical path (HIP events on the stream it runs on);
`roofline.per_kernel` lists the others with their own bound; PMC-derived traffic / VALU figures come from profiles/
(tools/profile_round.sh).  `cpu_baseline` is the compiled reference itself (oracle/_ref/burst15, all host cores) on a
bounded sample of the same reads and database (its .acx is written from the device-built tables): differential wall time of
two sample sizes, which cancels its database load.  `parity_vs_reference`: the .b6 the reference wrote for that sample against
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


def db_paths(workdir, args):
    args.db_qlen = args.read_len + max(10, args.read_len // 10)
    tag = "b%d_v%d_l%d_q%d_i%s_k%d" % (args.n_base, args.n_variants, args.ref_len, args.db_qlen, args.id, args.K)
    return (os.path.join(workdir, "refs_%s.fa" % tag), os.path.join(workdir, "db_%s.edx" % tag), os.path.join(workdir, "db_%s.acx" % tag))


def build_db(workdir, args, rank=0):
    """rank 0 writes the shared database (args: read_len, n_base, n_variants, ref_len, variant_rate, id, K); returns (refs, edx, acx, done marker)"""
    from burst_amd import host
    os.makedirs(workdir, exist_ok=True)
    refs, edx, acx = db_paths(workdir, args)
    done = edx + ".done"
    if rank == 0 and not os.path.exists(done):
        t = time.time()
        host.synth_refs(refs, args.n_base, args.n_variants, args.ref_len, args.variant_rate, 7)
        t1 = time.time()
        db = host.Db.from_fasta(refs, args.db_qlen, args.id, shear_len=500)
        t2 = time.time()
        db.write(edx, None, db_qlen=args.db_qlen, thres=args.id)
        db.close()
        open(done, "w").write("ok")
        log("[bench] database built in %.1f s (references %.1f s, clumps %.1f s, .edx %.1f s); the accelerator is built on the device" % (time.time() - t, t1 - t, t2 - t1, time.time() - t2))
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
    from burst_amd import host
    refs, edx, acx, done = build_db(workdir, args, rank)
    edits = [int(x) for x in args.edits.split(",")]
    n_pool_reads = args.reads * args.pool
    reads_fa = os.path.join(workdir, "reads_%d_l%d_e%s_u%s_f%d.fa" % (n_pool_reads, args.read_len, "-".join(map(str, edits)), args.iupac, int(args.fr)))
    if rank == 0 and not os.path.exists(reads_fa + ".done"):
        t = time.time()
        host.synth_reads(refs, reads_fa, n_pool_reads, args.read_len, edits, rc=args.fr, iupac=args.iupac, seed=42)
        open(reads_fa + ".done", "w").write("ok")
        log("[bench] %d reads written in %.1f s" % (n_pool_reads, time.time() - t))
    if rank == 0 and getattr(args, "drop_refs", False) and os.path.exists(refs):
        os.remove(refs)          # very large databases: the FASTA (as large as the database in bases) is not needed once reads and .edx exist
    return refs, edx, acx, reads_fa, done


def cpu_baseline(edx, acx, reads_fa, args):
    """the compiled reference on the host cores: differential timing of two sample sizes cancels its DB load time"""
    exe = os.path.join(ROOT, "oracle", "_ref", "burst%d" % args.K)
    if args.no_cpu_baseline or not os.path.exists(exe):
        return None
    cores = os.cpu_count() or 1
    n1, n2 = args.cpu_sample // 6, args.cpu_sample
    tmp = os.path.dirname(reads_fa)
    times = []
    for n in (n1, n2):
        sample = os.path.join(tmp, "cpu_sample_%d.fa" % n)
        with open(reads_fa, "rb") as f, open(sample, "wb") as o:
            for _ in range(2 * n):
                o.write(f.readline())
        t = time.time()
        r = subprocess.run([exe, "-r", edx, "-a", acx, "-q", sample, "-o", sample + ".b6", "-m", args.mode, "-i", str(args.id),
                            "-t", str(cores), "--noprogress"] + (["-fr"] if args.fr else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            log("[bench] reference failed:", r.stdout[-400:])
            return None
        times.append(time.time() - t)
    dt = max(times[1] - times[0], 1e-6)
    cpu_baseline.sample_fa, cpu_baseline.sample_b6 = sample, sample + ".b6"        # the larger sample: parity_vs_reference compares with it
    return {"value": (n2 - n1) / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
            "sample": "oracle/_ref/burst%d (reference compiled with gcc -O3 -march=x86-64-v3 -fopenmp) -t %d, same .edx/.acx, "
                      "-m %s -i %s; differential wall time of the first %d vs %d reads of the pool (%.2f s vs %.2f s) = its align "
                      "phase incl. parse/sort/output of the extra reads, database load cancelled"
                      % (args.K, cores, args.mode, args.id, n1, n2, times[0], times[1])}


def end_to_end(edx, reads_fa, args, device):
    """the whole command line on the read pool: burst_hip -r DB.edx -ad -q reads -o out.b6 (database read, accelerator built on the
    device, query ingest beside it, search, consolidation, .b6 written) -- its own 'Alignment time' and phase lines"""
    import re
    out = reads_fa + ".e2e.b6"
    cmd = [os.path.join(ROOT, "burst_amd", "burst_hip"), "-r", edx, "-ad", "-k", str(args.K), "-q", reads_fa, "-o", out, "-m", args.mode, "-i", str(args.id), "--device", str(device)] + (["-fr"] if args.fr else [])
    best = None
    for _ in range(2):      # the second run has the files in the page cache, as the bench's own inputs are
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        m = re.search(r"Alignment time: ([\d.]+) seconds", r.stdout)
        if r.returncode != 0 or not m:
            return {"error": r.stdout[-400:]}
        n = int(re.search(r"Parsed (\d+) queries", r.stdout).group(1))
        best = {"reads": n, "seconds": float(m.group(1)), "reads_per_s": n / float(m.group(1)), "lines": int(re.search(r"Wrote (\d+) alignments", r.stdout).group(1)),
                "phases_s": {k.strip(): float(v) for k, v in re.findall(r"\[([a-z ,()+.]+?)\s+([\d.]+) s", r.stdout)},
                "command": "burst_hip -r DB.edx -ad -k %d -q <%d reads> -o out.b6 -m %s -i %s (second of two runs)" % (args.K, n, args.mode, args.id)}
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


def pmc_table():
    """per-kernel PMC figures of the last profiling pass kept under profiles/ (tools/profile_round.sh)"""
    for name in ("pmc_summary.json", "traffic.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            try:
                return json.load(open(f))
            except Exception:
                pass
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=2000000, help="reads per step (per rank with --scaling weak, all GPUs together with --scaling strong)")
    ap.add_argument("--pool", type=int, default=4, help="distinct batches the steps cycle through")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--db-scale", type=float, default=1.0, help="multiplies --n-base (1 = 3.2 M references / 4.5 Gbp)")
    ap.add_argument("--n-base", type=int, default=1600000, help="random base sequences (round 2's family-dominated database: --n-base 33000 --n-variants 30)")
    ap.add_argument("--n-variants", type=int, default=2)
    ap.add_argument("--ref-len", type=int, default=1400)
    ap.add_argument("--variant-rate", type=float, default=0.05)
    ap.add_argument("--id", type=float, default=0.98)
    ap.add_argument("--K", type=int, default=15, choices=[12, 15], help="accelerator word length (the reference's DB12 / DB15 builds)")
    ap.add_argument("--mode", default="BEST")
    ap.add_argument("--workdir", default=os.environ.get("BURST_BENCH_DIR", "/tmp/burst_amd_bench"))
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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: weak = every rank aligns --steps batches of --reads reads (the job grows with N; default: the path partitions and "
                    "has no data-path collective), strong = the single-GPU job of --steps batches cut into N equal shares")
    ap.add_argument("--gather", default="shm", choices=["shm", "rccl"], help="N > 1: how the ranks' records reach rank 0 -- shared-memory segments rank 0 maps (default; no collective) or the library's RCCL gather")
    ap.add_argument("--acx-file", action="store_true", help="round 2's path: the accelerator from an .acx file (built by the host builder) instead of the device build")
    ap.add_argument("--ab", action="append", default=[], help="N = 1: after the timed region, time the same steps again with these library options (name=value[,name=value]; repeatable) "
                    "on the same resident database -- extra key `ab` of the JSON line, an A/B on one box in one process")
    args = ap.parse_args()
    args.n_base = int(round(args.n_base * args.db_scale))

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
        if one_dev is not None:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from burst_amd import capi, host
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
    def job_share(first, count, r=rank):
        # weak scaling (default; the path partitions, every rank does what the single GPU does): rank r aligns `count` pool batches
        # of its own, starting r batches further into the pool so that the ranks are not on the same reads at the same time.
        # strong scaling: the single-GPU job cut into N equal shares
        if weak:
            return [pool_range((first + k + r) % P) for k in range(count)]
        return share_of_job([pool_range((first + k) % P) for k in range(count)], r, world)
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
    def search(ranges):
        """one job share through the product's scheduler; N > 1: + the gather of the records to rank 0"""
        if use_dist:
            return rs.search(qs, ranges, args.mode, batch_uniq)
        return host.align_ranges(dev, qs, ranges, args.mode, batch_uniq, run=_own)

    # warm-up: sizes the library's grow-only buffers for this workload and runs W untimed steps
    # (the page-locked record buffer is allocated once, outside the timed region: the command line does it once per job as well)
    _own = host.Run()
    ent_per_step = min(batch_uniq, max((b - a for a, b in job_share(0, P)), default=1)) * (2 if args.fr else 1)
    share = max(1, sum(b - a for a, b in job_share(args.warmup, args.steps))) * (2 if args.fr else 1)
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
    search(job_share(0, max(1, args.warmup)))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    run = search(job_share(args.warmup, args.steps))
    n_records = int(run.c.nHits)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.time() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=pdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nr = torch.tensor([n_records if rank == 0 else 0], dtype=torch.int64, device=pdev)
        dist.all_reduce(nr)
        n_records = int(nr.item())
    total_reads = sum(reads_per_pool_batch[(args.warmup + k) % P] for k in range(args.steps))
    if weak and world > 1:      # every rank its own `steps` batches
        total_reads = sum(reads_per_pool_batch[(args.warmup + k + r) % P] for r in range(world) for k in range(args.steps))

    if rank == 0 and os.environ.get("BHIP_PROF"):      # library built with EXTRA_HIPFLAGS=-DPFM_PROF: wave-cycles per prefilter phase
        import ctypes
        lib = capi.lib()
        if hasattr(lib, "bhip_debug_prof"):
            arr = (ctypes.c_ulonglong * 8)()
            lib.bhip_debug_prof(arr, 1)
            tot = float(sum(arr)) or 1.0
            log("[bench] prefilter phase share: " + " ".join("%d:%.1f%%" % (i, 100.0 * v / tot) for i, v in enumerate(arr)) + "  total wave-cycles %.3g" % tot)
    if rank == 0:
        st, nb, sec_align = rs.own_stats() if use_dist else (run.stats(), int(run.c.nBatches), float(run.c.secAlign))
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
            pf_name = "k_prefilter_cf" if st["prefilter_algo"] == 0 else "k_prefilter_mask"
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
        pmc = pmc_table()
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
                if v[2] == "hbm" and pk.get("hbm_bytes_per_launch_gather_calibrated"):      # sector gathers: profiles/r02k_fetch_calibration.txt
                    e["traffic_gather_calibrated"] = pk["hbm_bytes_per_launch_gather_calibrated"]
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
        # measured slope over database sizes (tools/slope_fit.py over the bench lines kept under profiles/): ms per batch of --reads reads
        # = a + b x accelerator records per read; the metric's database has ~63 G list entries
        extrap = {"metric_database": "31.5 GB RefSeq .edx", "this_edx_bytes": edx_bytes, "size_ratio": scale_to_metric,
                  "acx_entries_here": acx_entries, "acx_records_per_read_here": rec_per_read}
        try:
            fit = json.load(open(os.path.join(ROOT, "profiles", "r03_slope.json")))
            extrap["fit"] = fit
        except Exception:
            extrap["fit"] = None
        res = {
            "metric": "aligned reads/sec (node), %d-bp synthetic reads @%s id vs RefSeq stand-in .edx/.acx (DB%d), -m %s; %d GPU" % (args.read_len, args.id, args.K, args.mode, world),
            "value": total_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32 bit-vectors (u8 edit distances)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3] shape on %d GPU(s): %d synthetic %d-bp reads per step (0-2 edits), -m %s -i %s, vs %d references x %d bp "
                                   "(%.2f Gbp; %d clumps; .edx %.2f GB + DB%d .acx %.2f GB); every step stages its batch afresh through the product's batch scheduler"
                                   % (world, args.reads, args.read_len, args.mode, args.id, args.n_base * args.n_variants, args.ref_len,
                                      args.n_base * args.n_variants * args.ref_len / 1e9, db.c.numRclumps, edx_bytes / 1e9, args.K, acx_bytes / 1e9),
                       "parallelism": ("query-sharded x%d (%s, in device batches of up to %d), DB replicated; " % (world, "weak scaling: every rank aligns %d batches of its own, the job is N times the single-GPU job" % args.steps
                                                                                                                     if weak else "strong scaling: rank r aligns the r-th N-th of the single-GPU job's unique queries", batch_uniq)) +
                                      ("no collective on the data path: every rank's record buffer is a page-locked shared-memory segment (its batches' records land there over its own PCIe link, behind the "
                                       "batch), rank 0 has the segments mapped side by side and reads the records where they lie (bh_node.c inside bh_search_multi_ex; bh_report_view consumes such a view) -- no copy" if use_dist and args.gather == "shm" else
                                       "one RCCL gather of the hit records to rank 0 (bhip_comm_gather_hits inside bh_search_multi, the function burst_hip --gpus N --gather rccl runs)"),
                       "timed_region": "bh_align_ranges over %d batches (copies + device routing two batches ahead, seed lookups + profiles one batch ahead, alignment, records to host memory)%s" %
                                       (nb, (" + hand-over of all ranks' records to rank 0 (shared memory)" if args.gather == "shm" else " + RCCL gather") if use_dist else ""),
                       "extrapolation": extrap,
                       "device": info["name"], "n_cu": info["n_cu"]},
            "roofline": {"bound": "hbm" if bound_dom == "hbm" else "valu", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": pmc_of(dom).get("hbm_bytes_per_launch"),
                         "traffic_gather_calibrated": pmc_of(dom).get("hbm_bytes_per_launch_gather_calibrated") if bound_dom == "hbm" else None,
                         "note": "dominant kernel by time on the critical path (HIP events on its stream; k_seed_ranges works for the next batch beside the chain, throttled: off_critical_path in per_kernel); per_kernel gives each kernel's own bound: the prefilter and the re-scorer are "
                                 "bound by HBM/LDS latency of short gathers, the k_myers_* sweeps by integer VALU issue (valu_frac = issued VALU instructions x 2 cycles / peak, from the PMC pass; half-rate VOP3 forms count once). traffic follows the guide's 2 x FETCH_SIZE rule; traffic_gather_calibrated = FETCH_SIZE + WRITE_SIZE, which is what a sector gather really moves (profiles/r02k_fetch_calibration.txt)",
                         "algorithmic_bytes_per_launch": bytes_dom, "ms_per_launch": ms_dom,
                         "per_kernel": per_kernel,
                         "gcups_sweeps": cells / (ms_sweeps * 1e-3) / 1e9 if ms_sweeps > 0 else 0.0},
            "phases_ms_per_batch": {k: per(k) for k in ("ms_h2d", "ms_peq", "ms_prefilter", "ms_seed", "ms_prefilter_hash", "ms_myers", "ms_myers_prefix", "ms_myers_window", "ms_rescore", "ms_d2h", "ms_total")},
            "work": {"records": n_records, "entries_per_batch": st["n_queries"] / nb, "raw_hits": st["n_raw_hits"], "hits": st["n_hits"],
                     "acx_entries_per_read": st["acx_entries_read"] / max(1.0, float(st["n_queries"])), "windows": st["n_windows"], "window_columns": st["n_window_columns"],
                     "lane_tasks_per_read": st["n_lane_tasks"] / max(1.0, float(st["n_queries"])), "task_columns": st["n_task_columns"],
                     "seed_words_per_read": st["n_seed_words"] / max(1.0, float(st["n_queries"]))},
            "host": {"db_read_s": t_db, "device_upload_s": t_dev, "query_ingest_s": t_q, "sec_in_device_calls": sec_align},
        }
        if world == 1 and args.ab:
            # A/B on the resident database: the same warm-up and steps under other tuning options, the default options' line again at the end
            defaults = {"prefilter_rb": 0, "seed_min_need": 3, "seed_drop_len": 8, "prefilter_table": 0, "prefilter_waves": 0, "prefilter_algo": -1, "prune": 1, "oversub": 2, "band": 1, "seed_ahead": 1}
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
        res["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "burst%d" % args.K)):      # N = 1 only (the contract); the other ranks would sit in the barrier meanwhile
            import shutil
            need_disk = acx_bytes + (2 << 30)
            if not os.path.exists(acx + ".done") and shutil.disk_usage(os.path.dirname(acx)).free < need_disk:
                log("[bench] no room for the reference's .acx (%.1f GB needed in %s): cpu_baseline skipped" % (need_disk / 1e9, os.path.dirname(acx)))
                res["cpu_baseline_skipped"] = "the reference reads an .acx file of %.1f GB; the work directory has %.1f GB free" % (acx_bytes / 1e9, shutil.disk_usage(os.path.dirname(acx)).free / 1e9)
            elif not os.path.exists(acx + ".done"):      # the reference reads an .acx FILE: written here from the tables the device built
                t = time.time()
                try:
                    db.acx_from_device(dev, args.K, 1)
                    host._chk(host.lib().bh_acx_write(C.byref(db.c), acx.encode()))
                    open(acx + ".done", "w").write("ok")
                    log("[bench] .acx for the reference written from the device-built tables in %.1f s (%.2f GB)" % (time.time() - t, os.path.getsize(acx) / 1e9))
                except Exception as e:
                    res["cpu_baseline_skipped"] = "could not write the reference's .acx: %s" % e
            if os.path.exists(acx + ".done"):
                try:
                    res["cpu_baseline"] = cpu_baseline(edx, acx, reads_fa, args)
                except Exception as e:
                    res["cpu_baseline_skipped"] = "reference run failed: %s" % e
        if world == 1 and not args.no_end_to_end:
            try:
                res["end_to_end"] = end_to_end(edx, reads_fa, args, local_rank)
            except Exception as e:      # reported, never fatal for the measurement
                res["end_to_end"] = {"error": str(e)}
            if res["end_to_end"] and "reads_per_s" in res["end_to_end"]:
                res["end_to_end_reads_per_s"] = res["end_to_end"]["reads_per_s"]
        if res["cpu_baseline"]:
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            try:
                res["parity_vs_reference"] = parity_vs_reference(dev, db, args, batch_uniq)
            except Exception as e:      # reported, never fatal for the measurement
                res["parity_vs_reference"] = {"error": str(e)}
        if use_dist:      # what rank 0 holds after the timed search, read once through (outside the timed region): every rank's run, its records
            runs = rs.view.runs()
            res["handover"] = {"kind": "view over the ranks' shared-memory segments" if (node is not None and rs.view.n_runs == world and world > 1) else "one array",
                               "records_per_run": [int(len(x)) for x in runs], "distinct_entries": int(sum(len(np.unique(x["q"])) for x in runs)),
                               "xor_of_reference_numbers": int(np.bitwise_xor.reduce(np.concatenate([x["refIx"] for x in runs]))) if sum(len(x) for x in runs) else 0}
        print(json.dumps(res), flush=True)
    if rs is not None:
        rs.close()
    _own.close()
    if comm:
        capi.lib().bhip_comm_destroy(comm)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
