"""Seeded synthetic references and reads (numpy), modelled on the behaviour of the reference's read
simulator embalmlets/LLsim.c:175-231: uniform start, fixed window length, an exact number of edits per read
(substitution : insertion : deletion = 3 : 1 : 1), optional reverse complement.  Symbols are the
reference's 4-bit codes (burst.c:1288-1307): A1 C2 G3 T4 N5 K6 M7 R8 Y9 S10 W11 B12 V13 H14 D15.
"""
import numpy as np

CODE2CHAR = np.frombuffer(b".ACGTNKMRYSWBVHD", dtype=np.uint8)
RC_CODE = np.array([0, 4, 3, 2, 1, 5, 7, 6, 9, 8, 10, 11, 13, 12, 15, 14], dtype=np.uint8)   # burst.c:168
# IUPAC codes compatible with each base (used to plant ambiguity codes that still match)
COMPAT = {1: [7, 8, 11, 13, 14, 15], 2: [7, 9, 10, 12, 13, 14], 3: [6, 8, 10, 12, 13, 15], 4: [6, 9, 11, 12, 14, 15]}


def random_genomes(n, length, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(1, 5, size=length, dtype=np.uint8) for _ in range(n)]


def mutate_family(base, n_variants, rate, rng):
    """variants of `base` with a fraction `rate` of edited positions (3:1:1 S:I:D)"""
    out = []
    for _ in range(n_variants):
        out.append(apply_edits(base, max(0, int(round(rate * len(base)))), rng))
    return out


def apply_edits(seq, n_edits, rng):
    seq = np.asarray(seq, dtype=np.uint8)
    if n_edits <= 0:
        return seq.copy()
    pos = np.sort(rng.choice(len(seq), size=min(n_edits, len(seq)), replace=False))
    kinds = rng.integers(0, 5, size=len(pos))          # 0-2 substitute, 3 delete, 4 insert
    out = []
    last = 0
    for p, k in zip(pos, kinds):
        out.append(seq[last:p])
        if k < 3:
            b = int(seq[p]) if 1 <= seq[p] <= 4 else 1
            out.append(np.array([(b - 1 + 1 + k) % 4 + 1], dtype=np.uint8))
            last = p + 1
        elif k == 3:
            last = p + 1
        else:
            out.append(np.array([rng.integers(1, 5)], dtype=np.uint8))
            last = p
    out.append(seq[last:])
    return np.concatenate(out)


def revcomp(codes):
    return RC_CODE[np.asarray(codes, dtype=np.uint8)[::-1]]


def make_reads(genomes, n_reads, read_len, n_edits, seed, rc_frac=0.0, iupac_frac=0.0):
    """n_edits: int or sequence of ints sampled uniformly per read.  Returns (reads, origin) with origin =
    (genome index, start, was_rc) per read."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(g) for g in genomes])
    ok = np.flatnonzero(lens > read_len)
    reads, origin = [], []
    choices = np.atleast_1d(np.asarray(n_edits))
    for _ in range(n_reads):
        gi = int(ok[rng.integers(0, len(ok))])
        st = int(rng.integers(0, lens[gi] - read_len))
        r = apply_edits(genomes[gi][st:st + read_len], int(choices[rng.integers(0, len(choices))]), rng)
        if iupac_frac > 0:
            m = np.flatnonzero((rng.random(len(r)) < iupac_frac) & (r >= 1) & (r <= 4))
            for p in m:
                opts = COMPAT[int(r[p])]
                r[p] = opts[rng.integers(0, len(opts))]
        rc = bool(rc_frac > 0 and rng.random() < rc_frac)
        if rc:
            r = revcomp(r)
        reads.append(r)
        origin.append((gi, st, rc))
    return reads, origin


def to_ascii(codes):
    return CODE2CHAR[np.asarray(codes, dtype=np.uint8)].tobytes().decode("ascii")


def write_fasta(path, seqs, names=None, prefix="s"):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%s\n%s\n" % (names[i] if names else "%s%d" % (prefix, i), to_ascii(s)))
