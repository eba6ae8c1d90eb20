"""Query sharding across the GPUs of one node: the database is replicated, the sorted unique
queries are cut into contiguous ranges (a query and its reverse-complement twin stay together, they share the
running minimum), every rank aligns its range independently, and ONE variable-length gather brings the 20-byte hit
records to rank 0, where the per-mode consolidation (incl. CAPITALIST's global vote, burst.c:4696-4727) runs.
torch.distributed only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
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
