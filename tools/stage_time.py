"""time bhip_stage_queries (H2D + host-side routing / seed plans) on the bench workload: python tools/stage_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from burst_amd import host
class A: pass
a = A(); a.read_len = 100; a.n_base = 3300; a.n_variants = 30; a.ref_len = 1400; a.variant_rate = 0.05; a.id = 0.97
a.K = 12
refs, edx, acx, done = bench.build_db("/tmp/burst_amd_bench", a)
bench.ensure_acx(edx, acx, 12)
reads = "/tmp/burst_amd_bench/st_reads.fa"
if not os.path.exists(reads):
    host.synth_reads(refs, reads, 1000000, 100, [0, 1, 2, 3], rc=False, iupac=0.0, seed=42)
db = host.Db.read(edx, acx, K=12)
qs = host.QuerySet(reads, 0.97, rc=False, accel=True, K=12)
dev = db.open_device(0)
q = qs.batch()
for i in range(4):
    t = time.time(); dev.stage(q); dt = time.time() - t
    print("stage %d: %.2f ms for %d entries (%d reads)" % (i, dt * 1e3, q.n, qs.n_reads))
