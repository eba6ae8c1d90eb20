import sys, os, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from burst_amd import host, capi
wd = "/tmp/burst_amd_bench"
tag = [f for f in os.listdir(wd) if f.endswith(".edx")][0][:-4]
edx, acx = os.path.join(wd, tag + ".edx"), os.path.join(wd, tag + ".acx")
reads = [f for f in os.listdir(wd) if f.startswith("reads_1000000")][0]
db = host.Db.read(edx, acx, K=12)
qs = host.QuerySet(os.path.join(wd, reads), 0.97, rc=False, accel=True, K=12)
n = qs.n_uniq
def mk(u0, u1):
    d = db.open_device(0); q = qs.batch(u0, u1); d.stage(q); d.align_staged(); d.align_staged(); return d
full = mk(0, n)
t = time.time(); [full.align_staged() for _ in range(5)]; tf = (time.time() - t) / 5
for parts in (2, 4):
    devs = [mk(n * i // parts, n * (i + 1) // parts) for i in range(parts)]
    t = time.time()
    for _ in range(5):
        for d in devs: d.align_staged()
    tseq = (time.time() - t) / 5
    def work(d):
        for _ in range(5): d.align_staged()
    th = [threading.Thread(target=work, args=(d,)) for d in devs]
    t = time.time(); [x.start() for x in th]; [x.join() for x in th]; tpar = (time.time() - t) / 5
    print("parts %d: full batch %.2f ms, sequential parts %.2f ms, concurrent streams %.2f ms" % (parts, tf * 1e3, tseq * 1e3, tpar * 1e3))
    for d in devs: d.close()
