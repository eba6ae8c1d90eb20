#!/usr/bin/env python3
"""bench.py -- aligned reads/sec of the BURST alignment hot path on MI355X.

Workload (BASELINE.json configs[1], the largest single-GPU configuration): 1 M synthetic 100-bp reads (0-3 edits,
LLsim-style) against a Greengenes-13.8-97%-like database (3 300 base sequences x 30 variants of 1.4 kb = 99 000
references / 139 Mbp, sheared at 500+113, K=12 accelerator), -m CAPITALIST -i 0.97.  Real Greengenes/RefSeq are
not reachable offline; sizes and generators are in DESIGN.md section 5.

A step = one pass of the whole hot path (profiles -> k-mer prefilter -> two-stage bit-parallel edit distance ->
re-scoring -> sorted hit records) over the batch, with the queries already resident in HBM (bhip_stage_queries) when
the timed region starts.  The records of a step reach host memory through the library's asynchronous hand-over: the
copy of step k runs while step k+1 computes, and all copies have landed before the clock stops (--sync-d2h copies
inside every step).  N > 1: one process per GPU (torch.distributed / RCCL), the database replicated, every rank aligns
its own shard of reads (weak scaling), and one padded gather of the 20-byte hit records per step brings them into
rank 0's HBM inside the timed region (device to device, asynchronous, double-buffered).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel by time (at the moment the lane-resolved
prefilter): algorithmic bytes per launch / HIP-event time of that launch;
`roofline.per_kernel` lists the sweeps as well.  `cpu_baseline` is the compiled reference itself (oracle/_ref/burst12,
all host cores) on a bounded sample, N = 1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(workdir, args, rank, world):
    """rank 0 writes the shared database; every rank writes its own reads"""
    from burst_amd import host
    os.makedirs(workdir, exist_ok=True)
    args.db_qlen = args.read_len + max(10, args.read_len // 10)
    K = getattr(args, "K", 12)
    tag = "b%d_v%d_l%d_q%d_i%s%s" % (args.n_base, args.n_variants, args.ref_len, args.db_qlen, args.id, "" if K == 12 else "_k%d" % K)
    refs = os.path.join(workdir, "refs_%s.fa" % tag)
    edx = os.path.join(workdir, "db_%s.edx" % tag)
    acx = os.path.join(workdir, "db_%s.acx" % tag)
    done = edx + ".done"
    if rank == 0 and not os.path.exists(done):
        t = time.time()
        host.synth_refs(refs, args.n_base, args.n_variants, args.ref_len, args.variant_rate, 7)
        db = host.Db.from_fasta(refs, args.db_qlen, args.id, shear_len=500, K=K)
        db.write(edx, acx, db_qlen=args.db_qlen, thres=args.id)
        db.close()
        open(done, "w").write("ok")
        log("[bench] database built in %.1f s" % (time.time() - t))
    return refs, edx, acx, done


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
    return {"value": (n2 - n1) / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
            "sample": "oracle/_ref/burst%d (reference compiled with gcc -O3 -march=x86-64-v3 -fopenmp) -t %d, same .edx/.acx, "
                      "-m %s -i %s; differential wall time of the first %d vs %d reads (%.2f s vs %.2f s) to cancel DB load"
                      % (args.K, cores, args.mode, args.id, n1, n2, times[0], times[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--n-base", type=int, default=3300)
    ap.add_argument("--n-variants", type=int, default=30)
    ap.add_argument("--ref-len", type=int, default=1400)
    ap.add_argument("--variant-rate", type=float, default=0.05)
    ap.add_argument("--id", type=float, default=0.97)
    ap.add_argument("--K", type=int, default=12, choices=[12, 15], help="accelerator word length (the reference's DB12 / DB15 builds)")
    ap.add_argument("--mode", default="CAPITALIST")
    ap.add_argument("--workdir", default=os.environ.get("BURST_BENCH_DIR", "/tmp/burst_amd_bench"))
    ap.add_argument("--cpu-sample", type=int, default=600000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fr", action="store_true", help="also search reverse complements (-fr)")
    ap.add_argument("--iupac", type=float, default=0.0, help="fraction of read bases replaced by a compatible IUPAC code")
    ap.add_argument("--edits", default="0,1,2,3", help="edit counts sampled per read")
    ap.add_argument("--one-stage", action="store_true", help="disable the prefix-filter stage of the edit-distance kernels")
    ap.add_argument("--lanes", type=int, default=0, help="sub-pipelines (HIP streams) per batch inside the library")
    ap.add_argument("--sweep-blocks", type=int, default=0, help="256-thread blocks per CU for the column sweep (0 = library default)")
    ap.add_argument("--sync-d2h", action="store_true", help="copy the records to the host inside every step (default: asynchronous hand-over, the copy of step k overlaps step k+1)")
    ap.add_argument("--opt", action="append", default=[], help="library tuning option name=value (bhip_set_option), repeatable")
    ap.add_argument("--prefilter-waves", type=int, default=0, help="single-wave prefilter blocks per CU (0 = library default)")
    ap.add_argument("--prefilter-stride", type=int, default=0, help="0 = automatic sparse seeds (default), 1 = every word (reference scheme)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    # BURST_BENCH_DIST1=1 (test hook, under torch.distributed.run with one process): take the N > 1 code path -- process group,
    # padded device-side gather, reductions -- on a single GPU
    use_dist = world > 1 or (os.environ.get("BURST_BENCH_DIST1") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from burst_amd import capi, host
    refs, edx, acx, done = build_inputs(args.workdir, args, rank, world)
    if use_dist:
        dist.barrier()
    while not os.path.exists(done):
        time.sleep(0.2)
    edits = [int(x) for x in args.edits.split(",")]
    reads_fa = os.path.join(args.workdir, "reads_%d_l%d_e%s_u%s_f%d_r%d.fa" % (args.reads, args.read_len, "-".join(map(str, edits)), args.iupac, int(args.fr), rank))
    if not os.path.exists(reads_fa):
        host.synth_reads(refs, reads_fa, args.reads, args.read_len, edits, rc=args.fr, iupac=args.iupac, seed=42 + rank)

    t = time.time()
    db = host.Db.read(edx, acx, K=args.K)
    qs = host.QuerySet(reads_fa, args.id, rc=args.fr, accel=True, K=args.K)
    dev = db.open_device(local_rank)
    dev.set_option("prefilter_stride", args.prefilter_stride)
    dev.set_option("two_stage", 0 if args.one_stage else 1)
    if args.lanes:
        dev.set_option("lanes", args.lanes)
    if args.sweep_blocks:
        dev.set_option("sweep_blocks", args.sweep_blocks)
    if args.prefilter_waves:
        dev.set_option("prefilter_waves", args.prefilter_waves)
    dev.set_option("async_d2h", 0 if args.sync_d2h else 1)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        dev.set_option(name, int(val))
    info = dev.info()
    q = qs.batch()
    dev.stage(q)
    log("[bench] rank %d: db %d refs / %d clumps, %d reads -> %d unique entries, load+stage %.1f s on %s"
        % (rank, db.c.totR, db.c.numRclumps, qs.n_reads, q.n, time.time() - t, info["name"]))
    all_hits = args.mode == "FORAGE"
    buf = None

    from burst_amd import dist as bdist

    # N > 1: every rank hands its records to its own host (as at N = 1) AND one RCCL gather per step brings all records into
    # rank 0's HBM, where they stay resident (burst_amd.dist.PaddedGather: device buffers filled by a device-to-device copy
    # from the library, asynchronous, double-buffered so the xGMI transfer overlaps the next step's alignment)
    pg = None

    bufs = [None, None]        # two host result buffers alternate: with the asynchronous hand-over the records of step k are
    turn = 0                   # still arriving while step k+1 runs (every copy is complete before the clock stops)

    def step():
        nonlocal turn
        hits, bufs[turn] = dev.align_staged(all_hits, bufs[turn])
        turn ^= 1
        if pg is not None:
            n = dev.copy_hits_device(pg.payload_ptr(), pg.cap)
            pg.post(n)
        return hits, None

    hits0, _ = step()          # sizes the library's grow-only device buffers for this workload (setup, not a warmup step)
    dev.sync_hits()
    if use_dist:            # same capacity on every rank: the largest shard's record count plus slack
        mx = torch.tensor([len(hits0)], dtype=torch.int64, device="cuda")
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        pg = bdist.PaddedGather(int(mx.item()) + int(mx.item()) // 8 + 4096, rank, world, torch.device("cuda", local_rank))
    for _ in range(args.warmup):
        step()
    dev.sync_hits()
    per_step = []
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        hits, _g = step()
        per_step.append(dev.stats(raw=True))
    dev.sync_hits()            # the last records are in host memory
    if pg is not None:
        pg.wait()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.time() - t0
    per_step = [s.as_dict() for s in per_step]
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nr = torch.tensor([qs.n_reads], dtype=torch.int64, device="cuda")
        dist.all_reduce(nr)
        total_reads = int(nr.item())
    else:
        total_reads = qs.n_reads

    if rank == 0 and os.environ.get("BHIP_PROF"):      # library built with EXTRA_HIPFLAGS=-DPFM_PROF: wave-cycles per prefilter phase
        import ctypes
        from burst_amd import capi as _capi
        lib = _capi.lib()
        if hasattr(lib, "bhip_debug_prof"):
            arr = (ctypes.c_ulonglong * 8)()
            lib.bhip_debug_prof(arr, 1)
            tot = float(sum(arr)) or 1.0
            log("[bench] prefilter phase share: " + " ".join("%d:%.1f%%" % (i, 100.0 * v / tot) for i, v in enumerate(arr)) + "  total wave-cycles %.3g" % tot)
    if rank == 0:
        st = per_step[-1]
        mean = lambda k: float(np.mean([s[k] for s in per_step]))
        two_stage = st["prefix_words"] > 0
        masked = st["prefilter_launches"] > 0
        # Algorithmic bytes per kernel launch (SURVEY.md 8d figures): prefilter = 8 B offset pair per
        # sampled word + 3 B per list entry (the SMALL .acx size) + 8 B per emitted task; column sweeps = 0.5 B (one 4-bit
        # symbol) per swept column of one reference lane + len/2 B of query and 12 B of result per unit.
        kernels = {}
        if masked and mean("ms_prefilter_hash") > 0:
            n = max(1, st["prefilter_launches"])
            pf_name = "k_prefilter_cf" if st["prefilter_algo"] == 0 else "k_prefilter_mask"
            kernels[pf_name] = (mean("ms_prefilter_hash") / n, (8.0 * st["n_seed_words"] + 3.0 * st["acx_entries_read"] + 8.0 * st["n_lane_tasks"]) / n)
        n = max(1, st["myers_launches"])
        if two_stage:
            cols = st["n_task_columns"] if masked else st["n_columns"] * 16
            units = st["n_lane_tasks"] if masked else st["n_pairs"] * 16
            kernels["k_myers_prefix%s<%d>" % ("_task" if masked else "", st["prefix_words"])] = (mean("ms_myers_prefix") / n, (0.5 * cols + units * (args.read_len / 2.0 + 12.0)) / n)
            kernels["k_myers_window<%d>" % ((args.read_len + 31) // 32)] = (mean("ms_myers_window") / n, (0.5 * st["n_window_columns"] + st["n_windows"] * (args.read_len / 2.0 + 12.0)) / n)
        else:
            kernels["k_myers<%d>" % ((args.read_len + 31) // 32)] = (mean("ms_myers") / n, st["bytes_algorithmic"] / n)
        dom = max(kernels, key=lambda k: kernels[k][0] * (st["prefilter_launches"] if k.startswith("k_prefilter") else n))
        ms_dom, bytes_dom = kernels[dom]
        achieved = bytes_dom / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(dom.split("<")[0], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        cells = (st["n_task_columns"] if masked else st["n_columns"] * 16.0) * min(args.read_len, 32.0 * max(1, st["prefix_words"])) + st["n_window_columns"] * float(args.read_len)
        ms_sweeps = mean("ms_myers")
        res = {
            "metric": "aligned reads/sec (node), %d-bp synthetic reads vs GG97-like .edx/.acx, -m %s -i %s" % (args.read_len, args.mode, args.id),
            "value": total_reads * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 bit-vectors (u8 edit distances)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic %d-bp reads per GPU vs GG97-like DB (%d refs x %d bp, %d clumps, K=%d .acx), -m %s -i %s"
                                   % (args.reads, args.read_len, args.n_base * args.n_variants, args.ref_len, db.c.numRclumps, args.K, args.mode, args.id),
                       "parallelism": "query-sharded x%d, DB replicated, RCCL gather of hit records" % world,
                       "device": info["name"], "n_cu": info["n_cu"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic,
                         "note": "dominant kernel by time; the prefilter is bound by HBM/LDS latency of short random list gathers, the k_myers_* sweeps by integer VALU issue (SURVEY 8d): their GCUPS is the truthful figure of merit",
                         "algorithmic_bytes_per_launch": bytes_dom, "ms_per_launch": ms_dom,
                         "per_kernel": {k: {"ms_per_launch": v[0], "algorithmic_bytes_per_launch": v[1], "GBps": (v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0)} for k, v in kernels.items()},
                         "gcups_sweeps": cells / (ms_sweeps * 1e-3) / 1e9 if ms_sweeps > 0 else 0.0},
            "phases_ms": {k: float(np.mean([s[k] for s in per_step])) for k in ("ms_peq", "ms_prefilter", "ms_seed", "ms_prefilter_hash", "ms_myers", "ms_myers_prefix", "ms_myers_window", "ms_rescore", "ms_d2h", "ms_total")},
            "work": {"pairs_per_read": st["n_pairs"] / max(1, q.n), "raw_hits": st["n_raw_hits"], "hits": st["n_hits"],
                     "acx_entries_per_read": st["acx_entries_read"] / max(1, q.n), "dp_columns": st["n_columns"],
                     "windows": st["n_windows"], "window_columns": st["n_window_columns"], "lane_tasks": st["n_lane_tasks"], "task_columns": st["n_task_columns"],
                     "seed_words_per_read": st["n_seed_words"] / max(1, q.n)},
        }
        res["cpu_baseline"] = cpu_baseline(edx, acx, reads_fa, args) if world == 1 else None      # N = 1 only (the contract); the other ranks would sit in the barrier meanwhile
        if res["cpu_baseline"]:
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
