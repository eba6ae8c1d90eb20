"""The N > 1 path on CPU: two gloo processes run the product's multi-rank search (bh_search_multi_ex through host.RankSearch, the
function behind burst_hip --gpus N, bench.py --gpus N and python -m burst_amd.run) with the ORACLE as the ranks' align back end
in place of the device scheduler: every rank's records land in its shared-memory segment (bh_node.c), rank 0 reports from the
segments where they lie (bh_report_view).  The .b6 must equal the reference's golden output -- this pins the shares of unique
queries, the entry numbers, the hand-over and the global CAPITALIST vote; the database-sharded variant adds the clump shards, the
per-query minimum over the ranks (the launcher's all_reduce as reduce_min), the filter and the (query, reference) order."""
import os
import subprocess
import sys

import numpy as np
import pytest

import goldenlib as gl

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from burst_amd import capi, host
import oraclelib as ol
edx, qfa, out, mode, ident, fr, shard_db = sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
db = host.Db.read(edx)
qs = host.QuerySet(qfa, ident, rc=bool(fr), accel=False)
lut = ol.score_lut(1)
c0, part = 0, db
if shard_db:
    c0, c1 = host.clump_shard(db, world, rank)
    part = db.slice(c0, c1)
clump_len = host._view(part.c.clumpLen, part.c.numRclumps, np.uint32)
packed = host._view(part.c.packed, part.c.packedWords * 16, np.uint8)
def align(ranges, mode_no):          # the rank's back end: the oracle on its ranges of unique queries (and its clumps)
    parts = []
    for u0, u1 in ranges:
        q = qs.batch(u0, u1)
        h = ol.search(packed, clump_len, part.c.totR, q.codes, q.off, q.emac.astype(np.uint32), q.six, q.rc, q.n_shared, lut, mode == "FORAGE")
        h = h.copy(); h["q"] = q.entry_index[h["q"]].astype(np.uint32)      # local entry -> global entry
        parts.append(h.view(capi.HIT_DTYPE))
    h = np.concatenate(parts) if parts else np.zeros(0, capi.HIT_DTYPE)
    return h[np.lexsort((h["refIx"], h["q"]))]
def reduce_min(a):
    t = torch.from_numpy(a)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
jt = torch.tensor([int.from_bytes(os.urandom(6), "little") if rank == 0 else 0], dtype=torch.int64)
dist.broadcast(jt, 0)
job = "t%x" % int(jt.item())
node = host.Node(job, rank, world, 200000) if rank == 0 else None
dist.barrier()
if rank != 0:
    node = host.Node(job, rank, world, 200000)
rs = host.RankSearch(None, rank, world, None, c0=c0, node=node, align=align, reduce_min=reduce_min if shard_db else None)
u0, u1 = (0, qs.n_uniq) if shard_db else host.shard_range(qs.n_uniq, world, rank)
rs.search(qs, [(u0, u1)], mode, 1 << 18, shard_db=world if shard_db else 0)
if rank == 0:
    assert int(rs.counts.sum()) == int(rs.view.total) and (shard_db or rs.view.n_runs == world)
    host.report_view(out, db, qs, rs.view, mode, host.REP_MERGED_LIST)
dist.barrier()
rs.close()
assert not [f for f in os.listdir("/dev/shm") if job in f] or rank != 0
dist.destroy_process_group()
'''


def _run_two_ranks(tmp_path, name, shard_db):
    import socket
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    out = str(tmp_path / "o.b6")
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(w), gl.ROOT, ref, q, out, c["mode"], c["id"], str(fr), str(int(shard_db))],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert sorted(open(out, "rb").read().splitlines()) == gl.golden_lines(c)


@pytest.mark.parametrize("name", ["dna_q100_capitalist_noacx_t1_fr", "quick_q292_best_fr"])
def test_two_rank_gloo_matches_reference(name, tmp_path):
    """query-sharded: the database replicated, every rank its share of the unique queries"""
    _run_two_ranks(tmp_path, name, False)


def test_shard_ranges_partition():
    from burst_amd import host
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = [host.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_job_shares_partition_the_job():
    """bench.py N > 1: rank r aligns the r-th N-th of the job (a sequence of pool batches): the shares are disjoint, in order,
    cover the job exactly and differ by at most one query"""
    import bench
    U, P = 7999135, 4
    pool = [(b * U // P, (b + 1) * U // P) for b in range(P)]
    for steps in (1, 3, 5, 20):
        job = [pool[(5 + k) % P] for k in range(steps)]
        flat = np.concatenate([np.arange(a, b) for a, b in job]) if steps < 5 else None
        total = sum(b - a for a, b in job)
        for world in (1, 2, 3, 8):
            shares = [bench.share_of_job(job, r, world) for r in range(world)]
            sizes = [sum(b - a for a, b in sh) for sh in shares]
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
            assert all(b > a for sh in shares for a, b in sh)
            if flat is not None:
                got = np.concatenate([np.arange(a, b) for sh in shares for a, b in sh])
                assert np.array_equal(got, flat)


def test_clump_shards_partition_the_database():
    """bh_clump_shard (--shard db): contiguous clump ranges that cover the database, about the same number of reference columns each"""
    from burst_amd import host
    db = host.Db.read(os.path.join(gl.G, "dna.edx"))
    cl = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32).astype(np.int64)
    for world in (1, 2, 3, 5, 8):
        prev, cols = 0, []
        for r in range(world):
            c0, c1 = host.clump_shard(db, world, r)
            assert c0 == prev and c1 >= c0
            cols.append(int(cl[c0:c1].sum()))
            prev = c1
        assert prev == db.c.numRclumps
        assert max(cols) - min(cols) <= 2 * int(cl.max())
    db.close()


@pytest.mark.parametrize("name", ["dna_q100_capitalist_noacx_t1_fr", "dna_q292_forage_noacx_t1_fr"])
def test_two_rank_database_sharding_matches_reference(name, tmp_path):
    """the second multi-GPU mode: the DATABASE is cut (bh_clump_shard, bh_db_slice), every rank searches all queries in its clumps
    (oracle in place of the device), the per-query minimum is combined over the ranks (not in FORAGE), what lies above it is dropped,
    rank 0 puts the records of both segments in (query, reference) order"""
    _run_two_ranks(tmp_path, name, True)


def test_database_slice_accelerator_equals_rebuilt():
    """bh_db_slice restricts every accelerator list to the slice's clumps: the result must be the accelerator bh_acx_build
    makes for the slice alone (same lengths, same packed lists, same BadList), and the slices partition the entries"""
    import ctypes as C
    import numpy as np
    from burst_amd import host
    db = host.Db.read(os.path.join(gl.G, "dna.edx"))
    host._chk(host.lib().bh_acx_build(C.byref(db.c), 12, 1))
    plain = host.Db.read(os.path.join(gl.G, "dna.edx"))
    cl = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    total = 0
    for world in (1, 3):
        total = 0
        for rank in range(world):
            c0, c1 = host.clump_shard(db, world, rank)
            a = db.slice(c0, c1)
            b = plain.slice(c0, c1)
            host._chk(host.lib().bh_acx_build(C.byref(b.c), 12, 1))
            la, lb = host._view(a.c.acxLens, 1 << 24, np.uint32), host._view(b.c.acxLens, 1 << 24, np.uint32)
            assert (la == lb).all() and a.c.acxListBytes == b.c.acxListBytes and a.c.acxFmt == b.c.acxFmt
            assert host._view(a.c.acxLists, a.c.acxListBytes, np.uint8).tobytes() == host._view(b.c.acxLists, b.c.acxListBytes, np.uint8).tobytes()
            assert a.c.badSz == b.c.badSz and a.c.totR == min(db.c.totR, 16 * c1) - 16 * c0
            total += int(la.sum(dtype=np.uint64))
        assert total == int(host._view(db.c.acxLens, 1 << 24, np.uint32).sum(dtype=np.uint64))


@pytest.mark.parametrize("n,n_entries,per_entry", [(0, 5, 1), (1, 5, 1), (1000, 300, 8), (300000, 90000, 6), (200000, 50, 9000)])
def test_order_records_equals_a_sort_by_entry_and_reference(n, n_entries, per_entry):
    """bh_order_records (the last step of a database-sharded search: the ranks' records into the order one device holding the
    whole database produces) against numpy's sort by (entry, reference): every field of every record, for a few and for many
    records per entry (FORAGE); below 65 536 records one thread works, above a team does"""
    import ctypes as C
    from burst_amd import capi, host
    L = host.lib()
    L.bh_order_records.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.bh_order_records.restype = C.c_int
    rng = np.random.default_rng(n + per_entry)
    # unique (entry, reference) pairs, then shuffled the way rank runs arrive: runs that are sorted inside, concatenated
    q = rng.integers(0, n_entries, size=n, dtype=np.uint32)
    ref = rng.integers(0, 1 << 22, size=n, dtype=np.uint32)
    key = np.unique(q.astype(np.uint64) << 32 | ref)
    h = np.zeros(len(key), dtype=capi.HIT_DTYPE)
    h["q"] = (key >> 32).astype(np.uint32); h["refIx"] = (key & 0xFFFFFFFF).astype(np.uint32)
    h["finalPos"] = rng.integers(0, 1 << 30, size=len(h)); h["score"] = rng.random(len(h), dtype=np.float32); h["ed"] = rng.integers(0, 255, size=len(h))
    want = h.copy()                                                   # np.unique sorted the keys
    ranks = rng.integers(0, 8, size=len(h))
    got = np.concatenate([h[ranks == r] for r in range(8)]) if len(h) else h.copy()
    a = np.ascontiguousarray(got.copy())
    assert L.bh_order_records(a.ctypes.data, len(a), n_entries) == 0
    assert a.tobytes() == want.tobytes()
    if len(h):
        bad = got.copy(); bad["q"][0] = n_entries
        assert L.bh_order_records(bad.ctypes.data, len(bad), n_entries) != 0 or len(bad) < 2


def test_minima_merge_is_the_elementwise_minimum():
    """bh_minima_merge (database-sharded ranks in one process: one byte per unique query, 255 = no hit on that rank)"""
    import ctypes as C
    from burst_amd import host
    L = host.lib()
    L.bh_minima_merge.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint64]
    L.bh_minima_merge.restype = None
    rng = np.random.default_rng(5)
    for n in (1, 70001, 300000):
        tabs = [rng.integers(0, 256, size=n, dtype=np.uint8) for _ in range(5)]
        want = np.minimum.reduce([tabs[0], tabs[1], tabs[3], tabs[4]])
        ptrs = (C.c_void_p * 5)(tabs[0].ctypes.data, tabs[1].ctypes.data, None, tabs[3].ctypes.data, tabs[4].ctypes.data)      # a NULL table is skipped
        L.bh_minima_merge(ptrs, 5, n)
        assert np.array_equal(tabs[0], want)


NODE_WORKER = r'''
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from burst_amd import capi, host
job, rank, world, cap, mode = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
L = host.lib()
node = host.Node(job, rank, world, cap)
run, allr = host.BhRun(), host.BhRun()
def records(r, call, n):
    h = np.zeros(n, dtype=capi.HIT_DTYPE)
    h["q"] = np.arange(n, dtype=np.uint32) + 1000003 * r; h["refIx"] = call; h["finalPos"] = r; h["score"] = np.float32(0.5) * call
    return h
def count(r, call):
    return [0, 17, cap, 3 * cap + 5, 1, cap // 2][call % 6] + r      # empty, small, full, beyond the allocated part (private buffer), ...
for call in range(1, 8):
    host._chk(L.bh_node_begin(node.h))
    L.bh_node_attach(node.h, C.byref(run))
    n = count(rank, call)
    keep = None
    if n > run.capHits:                     # what bh_align_ranges does when a search outgrows the buffer: a private, larger one
        keep = records(rank, call, n); run.hits = keep.ctypes.data; run.capHits = n; run.hitsPinned = 2
    else:
        src = records(rank, call, n); C.memmove(run.hits, src.ctypes.data, n * 20)
    run.nHits = n
    status = 7 if (mode == "fail" and rank == 1 and call == 3) else 0
    if mode == "late" and rank == 1 and call == 2:
        time.sleep(1.0)
    if mode == "dead" and rank == 1 and call == 2:
        sys.exit(0)                         # never publishes
    host._chk(L.bh_node_publish(node.h, C.byref(run), status))
    if keep is not None:
        run.hits = None; run.capHits = 0; run.hitsPinned = 0
    if rank == 0:
        counts = np.zeros(world, np.uint64)
        view = host.BhRunView()
        rc = L.bh_node_collect_view(node.h, C.byref(view), counts.ctypes.data_as(host.u64p)) if mode != "copy" else -6
        viewed = rc == 0
        if rc == -6:                        # BH_E_CAPACITY: a rank's records are not inside its slot (it outgrew its segment): one copy
            assert mode == "copy" or max(count(r, call) for r in range(world)) > cap, (call, rc)
            rc = L.bh_node_collect(node.h, C.byref(allr), counts.ctypes.data_as(host.u64p))
        if mode == "fail" and call == 3:
            assert rc != 0 and b"rank 1 failed" in L.bh_last_error(), L.bh_last_error()
            continue
        if mode == "dead" and call == 2:
            assert rc != 0 and b"rank 1 did not deliver" in L.bh_last_error(), L.bh_last_error()
            print("NODE_OK"); sys.exit(0)
        assert rc == 0, L.bh_last_error()
        want = np.concatenate([records(r, call, count(r, call)) for r in range(world)])
        assert [int(c) for c in counts] == [count(r, call) for r in range(world)]
        if viewed:
            assert max(count(r, call) for r in range(world)) <= cap and view.total == len(want) and view.n_runs == world
            got = np.concatenate(view.runs())
        else:
            got = np.frombuffer((C.c_uint8 * (int(allr.nHits) * 20)).from_address(allr.hits), dtype=capi.HIT_DTYPE) if allr.nHits else np.zeros(0, capi.HIT_DTYPE)
        assert got.tobytes() == want.tobytes(), (call, len(got), len(want))
run.hits = None
node.close()
print("NODE_OK")
'''


@pytest.mark.parametrize("mode,world", [("plain", 3), ("copy", 2), ("late", 2), ("fail", 2), ("dead", 2)])
def test_node_exchange_between_processes(mode, world, tmp_path):
    """bh_node.c, the hand-over of the records between the processes of one node (bench.py --gpus N, one process per GPU): every
    rank's records lie in its shared-memory segment; rank 0 reads them where they lie (a view over the segments, mapped side by
    side) or -- "copy", and whenever a rank outgrew its segment -- concatenates them in rank order.  Seven searches in a row with empty,
    small, full and outgrown buffers; a rank that is late; a rank that fails (rank 0 says which); a rank that dies (rank 0 gives
    up after BURST_NODE_TIMEOUT instead of waiting for ever); no segment is left in /dev/shm"""
    job = "t%d%s" % (os.getpid(), mode)
    env = dict(os.environ, BURST_NODE_TIMEOUT="10" if mode == "dead" else "90")
    ps = [subprocess.Popen([sys.executable, "-c", NODE_WORKER, gl.ROOT, job, str(r), str(world), "5000", mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for r in range(world)]
    outs = [p.communicate(timeout=120) for p in ps]
    assert "NODE_OK" in outs[0][0], outs[0]
    for r, p in enumerate(ps):
        assert p.returncode == 0, outs[r]
    left = [f for f in os.listdir("/dev/shm") if job in f]
    if mode != "dead":      # (the rank that died could not unlink its segment: the next job of that name does)
        assert not left, left
    for f in left:
        os.unlink(os.path.join("/dev/shm", f))


@pytest.mark.parametrize("name", ["dna_q100_capitalist_noacx_t1_fr", "quick_q292_best_fr"])
def test_report_from_runs_equals_report_from_one_array(name, tmp_path):
    """bh_report_view: the records of a search lying in several runs of one address range with something else between them (what
    rank 0 sees of a node's ranks: their shared-memory segments side by side) give the same .b6, byte for byte, as the same
    records in one array -- and that is the reference's golden output.  Records from the oracle, three ranks' shares, the runs in
    a buffer whose gaps hold records that must not be read (entry numbers beyond the job)."""
    import ctypes as C
    from burst_amd import capi, host
    import oraclelib as ol
    c = [x for x in gl.cases() if x["name"] == name][0]
    ref, q, fr, z, shear = gl.case_args(c)
    db = host.Db.read(ref)
    qs = host.QuerySet(q, float(c["id"]), rc=bool(fr), accel=False)
    lut = ol.score_lut(1)
    clump_len = host._view(db.c.clumpLen, db.c.numRclumps, np.uint32)
    packed = host._view(db.c.packed, db.c.packedWords * 16, np.uint8)
    shares = []
    for r in range(3):
        u0, u1 = host.shard_range(qs.n_uniq, 3, r)
        b = qs.batch(u0, u1)
        h = ol.search(packed, clump_len, db.c.totR, b.codes, b.off, b.emac.astype(np.uint32), b.six, b.rc, b.n_shared, lut, c["mode"] == "FORAGE")
        h = h.copy(); h["q"] = b.entry_index[h["q"]].astype(np.uint32)
        shares.append(h.view(capi.HIT_DTYPE))
    one = str(tmp_path / "one.b6"); runs = str(tmp_path / "runs.b6")
    host.report(one, db, qs, np.concatenate(shares), c["mode"], host.REP_MERGED_LIST)
    gap = 1000
    buf = np.zeros(sum(len(s) for s in shares) + 4 * gap, dtype=capi.HIT_DTYPE)
    buf["q"] = 0xFFFFFFFF
    v = host.BhRunView(); v.base = buf.ctypes.data; v.n_runs = 3
    at = gap
    for r, s in enumerate(shares):
        buf[at:at + len(s)] = s
        v.off[r] = at; v.n[r] = len(s); at += len(s) + gap
    v.total = sum(len(s) for s in shares)
    host.report_view(runs, db, qs, v, c["mode"], host.REP_MERGED_LIST)
    assert open(one, "rb").read() == open(runs, "rb").read()
    assert sorted(open(runs, "rb").read().splitlines()) == gl.golden_lines(c)


HANDOVER_HARNESS = r'''
import os, sys, textwrap, types
sys.path.insert(0, sys.argv[2])
import torch, torch.distributed as dist
from burst_amd import host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
src = open(os.path.join(sys.argv[2], "bench.py")).read()
a = src.index("            jt = torch.tensor([int.from_bytes(os.urandom(6)")
b = src.index('                args.gather = "rccl"\n') + len('                args.gather = "rccl"\n')
block = textwrap.dedent(src[a:b])
args = types.SimpleNamespace(gather="shm")
cap_rec, pdev, one_dev, node = 5000, "cpu", None, None
def log(m): print(m, flush=True)
mode = sys.argv[1]
if mode == "fail1" and rank == 1:
    real = host.Node
    def bad(*a, **k): raise host.HostError("simulated: no room in /dev/shm")
    host.Node = bad
if mode == "fail0" and rank == 0:
    def bad(*a, **k): raise host.HostError("simulated: no room in /dev/shm")
    host.Node = bad
g = dict(globals())
exec(block, g)
node, gather = g["node"], g["args"].gather
print("rank", rank, "mode", mode, "node", node is not None, "gather", gather, flush=True)
open(os.path.join(sys.argv[3], "result.%d" % rank), "w").write("%d %s" % (int(node is not None), gather))
assert (mode == "ok") == (node is not None) and gather == ("shm" if mode == "ok" else "rccl")
if node is not None: node.close()
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,port", [("ok", 29741), ("fail1", 29742), ("fail0", 29743)])
def test_bench_chooses_the_handover_together(mode, port, tmp_path):
    """bench.py, N > 1: the block that opens the ranks' shared-memory segments (run verbatim out of bench.py, two gloo processes):
    all ranks end with a segment, or -- when any rank cannot have one (simulated: /dev/shm full on rank 1 / on rank 0) -- all of them
    close what they opened and take the RCCL gather; nobody is left waiting and nothing stays in /dev/shm"""
    w = tmp_path / "harness.py"
    w.write_text(HANDOVER_HARNESS)
    import socket
    with socket.socket() as sk:          # a port nobody holds right now (the fixed one may still be in TIME_WAIT from an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port), str(w), mode, gl.ROOT,
                        str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    want = "1 shm" if mode == "ok" else "0 rccl"
    for rk in range(2):                  # (files, not the launcher's merged stdout: two processes' lines can interleave there)
        assert open(str(tmp_path / ("result.%d" % rk))).read() == want, r.stdout[-2000:]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("burst_hip.bench")]


SHARE_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import ctypes as C
import torch, torch.distributed as dist
from burst_amd import host
edx, K, fail_rank = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
# the tables of the single-rank build: the host builder's Lens and list area (what every rank must hold in the end)
db = host.Db.read(edx)
host._chk(host.lib().bh_acx_build(C.byref(db.c), K, 1))
nw = 1 << (2 * K)
lens = host._view(db.c.acxLens, nw, np.uint32).copy()
lists = host._view(db.c.acxLists, db.c.acxListBytes, np.uint8).copy()
per = {0: 5, 1: 3}[int(db.c.acxFmt)]          # bytes of a pair of entries (SMALL) / of an entry (LARGE)
# the ranks' runs of words: equal numbers of entries, on word boundaries (as the device builder cuts them from its histogram)
cum = np.concatenate(([0], np.cumsum(lens, dtype=np.uint64)))
cuts = [0] + [int(np.searchsorted(cum, cum[-1] * r // world)) for r in range(1, world)] + [nw]
if int(db.c.acxFmt) == 0:                  # (SMALL packs two entries into five bytes: regions on even entry numbers)
    cuts = [0] + [int(c) for c in cuts[1:-1] if cum[c] % 2 == 0] + [nw]
    cuts += [nw] * (world + 1 - len(cuts))
def entry_bytes(w):
    e = int(cum[w]); return e * 3 if per == 3 else (e // 2) * 5 + (e % 2) * 3
loff = [c * 4 for c in cuts]
roff = [entry_bytes(c) for c in cuts[:-1]] + [len(lists)]
# a rank holds ITS region only (the rest is junk), as after the device builder's own part
rng = np.random.default_rng(rank + 7)
def own_only(full, off):
    a = rng.integers(0, 256, len(full), dtype=np.uint8)
    a[off[rank]:off[rank + 1]] = full[off[rank]:off[rank + 1]]
    return a
my_lens, my_lists = own_only(lens.view(np.uint8), loff), own_only(lists, roff)
def any_failed(status):
    t = torch.tensor([1 if status else 0]); dist.all_reduce(t, op=dist.ReduceOp.MAX); return bool(int(t.item()))
def broadcast(buf, root):
    t = torch.from_numpy(buf); dist.broadcast(t, root); return t.numpy()
calls = [0]
def share(arr, off, status=0):
    def fetch(a, n): return arr[a:a + n].copy()
    def store(a, buf):
        calls[0] += 1
        if fail_rank >= 100 and rank == fail_rank - 100 and calls[0] == 2: raise RuntimeError("device copy failed (test)")
        arr[a:a + len(buf)] = buf
    return host.share_regions(off, rank, world, status, fetch, store, broadcast, any_failed, piece=70001)
st = share(my_lens, loff, 1 if rank == fail_rank else 0)
if fail_rank >= 100:
    assert st == -1          # a rank failed INSIDE the exchange: every rank walked through all broadcasts and learnt it at the end (no hang)
elif fail_rank >= 0:
    assert st == 1 and not np.array_equal(my_lens, lens.view(np.uint8))      # announced: nothing moved, on every rank
else:
    assert st == 0 and np.array_equal(my_lens, lens.view(np.uint8))
    assert share(my_lists, roff) == 0 and np.array_equal(my_lists, lists)
print("rank %d ok: %d words, %d list bytes, regions %s" % (rank, nw, len(lists), roff), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,K,fail_rank,port", [(2, 10, -1, 29751), (3, 8, -1, 29752), (2, 10, 1, 29753), (3, 10, 101, 29754)])
def test_cooperative_build_exchange_over_gloo(world, K, fail_rank, port, tmp_path):
    """The exchange of the cooperative accelerator build (bhip_share_fn) as python -m burst_amd.run does it for ranks without an RCCL
    communicator: host.share_regions over the launcher's process group.  Every rank holds the single-rank tables (the host builder's Lens
    and list area) in its own run of words only; after the two exchanges -- list lengths, then the lists -- every rank holds all of them.
    A rank that announces a failure makes the exchange return 1 everywhere with nothing moved; a rank whose copy fails in the MIDDLE
    (fail_rank = 100 + rank) keeps taking part and all ranks return -1 together.  (The device builder itself: -m gpu,
    tests/test_gpu_acx.py::test_cooperative_build_equals_the_single_rank_build.)"""
    w = tmp_path / "w.py"
    w.write_text(SHARE_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port), str(w),
                        gl.ROOT, os.path.join(gl.G, "quick.edx"), str(K), str(fail_rank)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count(" ok: ") == world, r.stdout[-3000:]
