"""numpy builders of the on-disk layouts (for unit tests that do not go through the C host):
.edx clump area (burst.c:2707-2737 clump build + 2810-2824 4-bit pack) and .acx tables (burst.c:3501-3530)."""
import numpy as np


def pack_clumps(seqs):
    """seqs: list of uint8 code arrays, taken 16 at a time in the given order.
    Returns (packed uint8[...], clump_len uint32[n_clumps], tot_refs)."""
    tot = len(seqs)
    n_clumps = (tot + 15) // 16
    clump_len = np.zeros(n_clumps, np.uint32)
    parts = []
    for c in range(n_clumps):
        group = seqs[16 * c:16 * c + 16]
        L = max(len(s) for s in group)
        clump_len[c] = L
        rows = np.zeros((L + (L & 1), 16), np.uint8)
        for z, s in enumerate(group):
            rows[:len(s), z] = s
        parts.append((rows[0::2] | (rows[1::2] << 4)).astype(np.uint8).reshape(-1))
    return np.concatenate(parts), clump_len, tot


def clump_rows(seqs, c):
    group = seqs[16 * c:16 * c + 16]
    L = max(len(s) for s in group)
    rows = np.zeros((L, 16), np.uint8)
    for z, s in enumerate(group):
        rows[:len(s), z] = s
    return rows


def build_acx(seqs, K, skip_clumps=()):
    """Word -> sorted unique clump ids for unambiguous words (burst.c:3378-3388).  Returns
    (lens uint32[4^K], entries uint32[...], offs uint64[4^K+1]); words containing a code outside 1..4 are skipped."""
    nw = 1 << (2 * K)
    pairs = []
    for i, s in enumerate(seqs):
        s = np.asarray(s, np.int64)
        if len(s) < K or (i // 16) in skip_clumps:
            continue
        ok = (s >= 1) & (s <= 4)
        v = np.where(ok, s - 1, 0)
        w = np.zeros(len(s) - K + 1, np.int64)
        good = np.ones(len(s) - K + 1, bool)
        for k in range(K):
            w = (w << 2) | v[k:len(s) - K + 1 + k]
            good &= ok[k:len(s) - K + 1 + k]
        w = w[good]
        pairs.append(np.stack([w, np.full(len(w), i // 16, np.int64)], 1))
    allp = np.unique(np.concatenate(pairs), axis=0) if pairs else np.zeros((0, 2), np.int64)
    lens = np.bincount(allp[:, 0], minlength=nw).astype(np.uint32)
    offs = np.zeros(nw + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    return lens, allp[:, 1].astype(np.uint32), offs


def pack_acx_lists(lens, entries, fmt):
    """fmt 0: SMALL (pairs of 20-bit ids in 5 bytes, odd tail 3 bytes, burst.c:3516-3527); fmt 1: LARGE (3 bytes each)"""
    out = bytearray()
    pos = 0
    if fmt == 1:
        for e in entries:
            out += int(e).to_bytes(3, "little")
        return np.frombuffer(bytes(out), np.uint8)
    nz = np.flatnonzero(lens)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))]).astype(np.int64)
    for w in nz:
        lst = entries[offs[w]:offs[w + 1]]
        for i in range(0, len(lst) - 1, 2):
            out += (int(lst[i]) | (int(lst[i + 1]) << 20)).to_bytes(5, "little")
        if len(lst) & 1:
            out += int(lst[-1]).to_bytes(3, "little")
    return np.frombuffer(bytes(out) + b"\0" * 8, np.uint8)
