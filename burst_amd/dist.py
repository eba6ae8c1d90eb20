"""Query sharding across the GPUs of one node: the database is replicated, the sorted unique
queries are cut into contiguous ranges (a query and its reverse-complement twin stay together, they share the
running minimum), every rank aligns its range independently, and ONE variable-length gather brings the 20-byte hit
records to rank 0, where the per-mode consolidation (incl. CAPITALIST's global vote, burst.c:4696-4727) runs.
torch.distributed only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

Second mode, for databases that do not fit one device next to their accelerator (capacity caveat):
the DATABASE is cut into contiguous clump ranges, every rank aligns ALL queries against its range, and because "the
hits of a query" are the references at the query's GLOBAL minimum edit distance (burst.c:4217-4277) the ranks exchange
that minimum -- one all_reduce(MIN) over a byte per unique query -- drop what lies above it, and then gather as before."""
import numpy as np

from . import capi


def shard_range(n_uniq, world, rank):
    base, rem = divmod(n_uniq, world)
    u0 = rank * base + min(rank, rem)
    return u0, u0 + base + (1 if rank < rem else 0)


def gather_hits(hits, rank, world, device="cpu", dst=0):
    """all_gather of record counts + one padded gather of the byte payload; returns the concatenation on dst (rank order)"""
    if world == 1:
        return hits
    import torch
    import torch.distributed as dist
    hits = np.ascontiguousarray(hits)
    cnt = torch.tensor([len(hits)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros(mx * capi.HIT_DTYPE.itemsize, dtype=torch.uint8, device=device)
    if len(hits):
        pad[:hits.nbytes] = torch.from_numpy(hits.view(np.uint8).reshape(-1)).to(device)
    out = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    parts = [o[:c * capi.HIT_DTYPE.itemsize].cpu().numpy().view(capi.HIT_DTYPE) for o, c in zip(out, counts)]
    return np.concatenate(parts) if parts else np.zeros(0, capi.HIT_DTYPE)


def run_sharded(n_uniq, align_range, rank, world, device="cpu"):
    """align_range(u0, u1) -> HIT_DTYPE records with q = GLOBAL entry index, records of one entry contiguous.
    Returns all records on rank 0 (None elsewhere)."""
    u0, u1 = shard_range(n_uniq, world, rank)
    hits = align_range(u0, u1) if u1 > u0 else np.zeros(0, capi.HIT_DTYPE)
    return gather_hits(hits, rank, world, device)


def clump_shard_range(clump_len, world, rank):
    """contiguous clump ranges with about the same number of reference columns each"""
    n = len(clump_len)
    cum = np.cumsum(np.asarray(clump_len, dtype=np.int64))
    total = int(cum[-1]) if n else 0
    cut = [0] + [int(np.searchsorted(cum, total * k / world, side="left")) for k in range(1, world)] + [n]
    for k in range(1, world + 1):
        cut[k] = max(cut[k], cut[k - 1])
    return cut[rank], cut[rank + 1]


def local_minimum(hits, shared_of_entry, n_shared):
    """smallest edit distance per shared query slot (a query and its reverse complement share one), 255 = no hit"""
    m = np.full(n_shared, 255, np.uint8)
    if len(hits):
        np.minimum.at(m, shared_of_entry[hits["q"]], hits["ed"])
    return m


def filter_minimum(hits, shared_of_entry, gmin):
    return hits[hits["ed"] == gmin[shared_of_entry[hits["q"]]]] if len(hits) else hits


def run_db_sharded(clump_len, shared_of_entry, n_shared, align_slice, rank, world, device="cpu", all_hits=False):
    """align_slice(c0, c1) -> HIT_DTYPE records of ALL queries against the clumps [c0, c1), q = global entry index and
    refIx = index in the whole database.  Returns on rank 0 the records a single device holding the whole database would
    have produced (same set, sorted by (q, refIx)); None elsewhere."""
    c0, c1 = clump_shard_range(clump_len, world, rank)
    hits = align_slice(c0, c1) if c1 > c0 else np.zeros(0, capi.HIT_DTYPE)
    if not all_hits and world > 1:
        import torch
        import torch.distributed as dist
        m = torch.from_numpy(local_minimum(hits, shared_of_entry, n_shared)).to(device)
        dist.all_reduce(m, op=dist.ReduceOp.MIN)
        hits = filter_minimum(hits, shared_of_entry, m.cpu().numpy())
    out = gather_hits(hits, rank, world, device)
    if out is None:
        return None
    return out[np.lexsort((out["refIx"], out["q"]))]
