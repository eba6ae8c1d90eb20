"""diagnostic: entries of the full-size workload without a hit -> are they found by the exhaustive route?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from burst_amd import host, capi
class A: pass
a = A(); a.read_len, a.n_base, a.n_variants, a.ref_len, a.variant_rate, a.id, a.K = 100, 3300, 30, 1400, 0.05, 0.97, 12
work = "/tmp/burst_amd_bench"
refs, edx, acx, done = bench.build_db(work, a)
bench.ensure_acx(edx, acx, 12)
reads = os.path.join(work, "fullsize_reads.fa")
if not os.path.exists(reads):
    host.synth_reads(refs, reads, 1000000, 100, [0, 1, 2, 3], rc=False, iupac=0.0, seed=4242)
db = host.Db.read(edx, acx, K=12)
qs = host.QuerySet(reads, 0.97, rc=False, accel=True, K=12)
dev = db.open_device(0)
q = qs.batch()
dev.stage(q)
base, _ = dev.align_staged(False)
found = np.zeros(q.n, bool); found[base["q"]] = True
miss = np.flatnonzero(~found)
print("entries", q.n, "missing", len(miss))
lens = np.diff(q.off.astype(np.int64))
print("length histogram of missing:", np.bincount(lens[miss])[90:110], "E of missing:", np.bincount(q.emac[miss]))
print("length histogram of all:", np.bincount(lens)[90:110])
sub = miss[:300]
seqs = [q.codes[int(q.off[i]):int(q.off[i + 1])] for i in sub]
q2 = capi.Queries(seqs, [int(q.emac[i]) for i in sub], list(range(len(sub))), [0] * len(sub))
q2.flags = np.full(q2.n, capi.BHIP_Q_EXHAUSTIVE, np.uint8)
h2 = dev.align_batch(q2)
print("exhaustive route finds hits for", len(np.unique(h2["q"])), "of", len(sub), "missing entries")
if len(h2):
    print(h2[:5])
