"""Multi-GPU front end: one process per GPU (torchrun), the alignment flags of the burst_hip command line (-r -a -q -o -m -i
-fr -y -k; the taxonomy flags -b*, -w, -t and direct-FASTA references are burst_hip only).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      -m burst_amd.run -r DB.edx -a DB.acx -q reads.fa -o out.b6 -m CAPITALIST -i 0.97 [-fr] [-y]

Every rank loads the database and the queries through the C host (libburst_host.so), aligns its contiguous shard of
unique queries with the C batch scheduler (bh_align_ranges -> libburst_hip.so); every rank's record buffer is a shared-memory
segment rank 0 has mapped (bh_node.c: no collective on the data path, the records cross each rank's own PCIe link behind its
batches), and rank 0 writes the .b6 with the C consolidation code straight from the segments (bh_report_view).  With one process
it is equivalent to burst_hip.
`--shard db` cuts the database instead of the queries (every rank aligns all queries against its clumps; one all_reduce(MIN) of the
per-query minimum -- the launcher's collective handed to bh_search_multi_ex as its reduce_min -- before the same hand-over) for
databases that do not fit one device."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="burst_amd.run")
    ap.add_argument("-r", "--references", required=True)
    ap.add_argument("-a", "--accelerator")
    ap.add_argument("-ad", "--accelerator-device", action="store_true", help="no .acx file: the accelerator is built on the devices from the database "
                    "(with several ranks and a replicated database: together, every rank the lists of its share of the words)")
    ap.add_argument("-q", "--queries", required=True)
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("-m", "--mode", default="CAPITALIST", choices=["BEST", "ALLPATHS", "CAPITALIST", "FORAGE", "ANY"])
    ap.add_argument("-i", "--id", type=float, default=0.97)
    ap.add_argument("-fr", "--forwardreverse", action="store_true")
    ap.add_argument("-y", "--nwildcard", action="store_true")
    ap.add_argument("-k", type=int, default=0, choices=[0, 12, 15], help="accelerator word length (0 = from the file's size)")
    ap.add_argument("--batch", type=int, default=1 << 21)
    ap.add_argument("--shard", default="queries", choices=["queries", "db"],
                    help="queries: database replicated, every rank aligns its range of queries (default); db: every rank holds a "
                         "range of the database's clumps and aligns all queries (for databases larger than one device)")
    args = ap.parse_args(argv)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch
    one_dev = os.environ.get("BURST_RUN_DEVICE")      # test hook: every rank on this device (gloo plumbing; RCCL refuses two ranks on one device)
    if one_dev is not None:
        local_rank = int(one_dev)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if one_dev is not None:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from burst_amd import host
    z = 0 if args.nwildcard else 1
    if args.accelerator_device and args.accelerator:
        sys.stderr.write("ERROR: -ad builds the accelerator on the device; drop -a\n")
        return 1
    db = host.Db.read(args.references, args.accelerator, K=args.k, z=z)
    K = int(db.c.K) if args.accelerator else (args.k or 12)
    accel = bool(args.accelerator or args.accelerator_device)
    host.lib().bh_queries_sort_device(local_rank)          # large query files are sorted on this rank's own device
    qs = host.QuerySet(args.queries, args.id, rc=args.forwardreverse, accel=accel, K=K, z=z)
    # the database was sheared for queries up to shear * id long: longer ones would lose alignments across shear boundaries
    # (burst.c:5152-5156: "DB incompatible with selected queries/identity", exit 1)
    if db.c.shear and int(np.float32(qs.c.maxLen) / np.float32(args.id)) > db.c.shear:
        sys.stderr.write("ERROR: DB incompatible with selected queries/identity.\n")
        return 1
    c0 = 0
    part = db
    shard_db = args.shard == "db" and world > 1
    if shard_db:
        c0, c1 = host.clump_shard(db, world, rank)
        part = db.slice(c0, c1) if c1 > c0 else None
    build_K = K if args.accelerator_device else 0
    if build_K and world > 1 and not shard_db and not os.environ.get("BURST_HIP_SOLO_BUILD"):
        # the replicated database's accelerator, built by the ranks together (bhip_build_accelerator_shared); the regions travel over the
        # launcher's process group: in place between the devices under the nccl back end (RCCL), through host memory under gloo
        share = host.dist_share(dist, "cpu" if one_dev is not None else "cuda")
        dev = db.open_device_shared(local_rank, z, build_K, rank, world, share)
    else:
        dev = part.open_device(local_rank, z, build_K=build_K) if part is not None else None
    if dev is not None:
        qs.pin()
    t0 = time.time()
    if world == 1:
        run = host.align_ranges(dev, qs, [(0, qs.n_uniq)], args.mode, args.batch)
        n = host.report(args.output, db, qs, run.hits, args.mode, 0 if accel else host.REP_MERGED_LIST)
        print("rank 0: %d hit records from 1 rank(s) in %.3f s, %d alignments written" % (int(run.c.nHits), time.time() - t0, n))
        return 0
    # several ranks, one process each: the C host's multi-rank search (bh_search_multi_ex), the function behind burst_hip --gpus N.
    # Query-sharded: rank r aligns the r-th share of the unique queries, no collective on the data path.  Database-sharded: every
    # rank aligns all queries against its clumps and the per-query minimum is combined over the ranks -- the launcher's own
    # all_reduce(MIN) (RCCL under the nccl backend) handed to the search as its reduce_min.  Either way every rank's record buffer is
    # a shared-memory segment rank 0 has mapped (bh_node.c) and rank 0 reports from there.
    pdev = "cpu" if one_dev is not None else "cuda"
    jt = torch.tensor([int.from_bytes(os.urandom(6), "little") if rank == 0 else 0], dtype=torch.int64, device=pdev)
    dist.broadcast(jt, 0)
    job = "run%x" % int(jt.item())
    u0, u1 = (0, qs.n_uniq) if shard_db else host.shard_range(qs.n_uniq, world, rank)
    strands = 2 if qs.n_entries > qs.n_uniq else 1
    cap = int((u1 - u0) * strands * (4.0 if args.mode in ("FORAGE", "ALLPATHS") else 1.5)) + (1 << 20)
    # the segments are opened TOGETHER: rank 0 first (the others map its segment), and if any rank cannot have one (/dev/shm too small)
    # every rank hears of it and the job ends with the reason on all of them instead of leaving the others in a barrier
    node, why = None, ""
    def try_open():
        try:
            return host.Node(job, rank, world, cap), ""
        except host.HostError as e:
            return None, str(e)
    if rank == 0:
        node, why = try_open()
    ok = torch.tensor([1 if (rank != 0 or node is not None) else 0], dtype=torch.int64, device=pdev)
    dist.broadcast(ok, 0)
    if int(ok.item()) and rank != 0:
        node, why = try_open()
    ok = torch.tensor([1 if (node is not None and (dev is not None or not shard_db)) else 0], dtype=torch.int64, device=pdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if not int(ok.item()):
        if node is not None:
            node.close()
        sys.stderr.write("rank %d: the ranks' shared-memory hand-over could not be set up%s\n" % (rank, ": " + why if why else " (another rank failed)"))
        dist.destroy_process_group()
        return 4
    def reduce_min(a):
        t = torch.from_numpy(a).to(pdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        a[:] = t.cpu().numpy()
    rs = host.RankSearch(dev, rank, world, None, c0=c0, node=node, reduce_min=reduce_min if shard_db else None)
    rs.search(qs, [(u0, u1)], args.mode, args.batch, shard_db=world if shard_db else 0)
    if rank == 0:
        n = host.report_view(args.output, db, qs, rs.view, args.mode, 0 if accel else host.REP_MERGED_LIST)
        print("rank 0: %d hit records from %d rank(s) in %.3f s, %d alignments written" % (int(rs.view.total), world, time.time() - t0, n))
    dist.barrier()      # (the ranks' segments live until rank 0 has written the report)
    rs.close()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
