"""Round 6 diagnosis: burst_hip at the metric's size started from a Python parent that held (and released) the same database on the
device runs out of device memory in bhip_reserve, started from a shell it does not.  Which ingredient matters?
  python tools/e2e_probe.py <workdir with the bench's db_*.edx and reads_*.fa>"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = sys.argv[1]
edx = sorted(glob.glob(os.path.join(W, "db_*.edx")))[0]
reads = sorted(glob.glob(os.path.join(W, "reads_*.fa")))[0]
cli = [os.path.join(ROOT, "burst_amd", "burst_hip"), "-r", edx, "-ad", "-k", "15", "-q", reads, "-o", os.path.join(W, "probe.b6"), "-m", "BEST", "-i", "0.98"]


def run(label, env=None):
    t = time.time()
    r = subprocess.run(cli, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, BHIP_DEBUG="1", **(env or {})))
    keep = [l for l in r.stdout.splitlines() if "after the build" in l or "[bhip] reserve" in l or "out of memory" in l or "Alignment time" in l or "references on the device" in l or "free at its start" in l or "query sort on the device" in l or "[bhip] ..." in l or "found no memory" in l]
    print("%-46s rc=%d %.1f s | %s" % (label, r.returncode, time.time() - t, " | ".join(x.strip()[:150] for x in keep)), flush=True)


run("A shell-like: no parent state")
import torch
torch.cuda.init(); torch.cuda.synchronize()
from burst_amd import host
db = host.Db.read(edx)
dev = db.open_device(0, build_K=15)
dev.close()
f_, t_ = torch.cuda.mem_get_info(0)
print("parent released the database: %.1f GB free" % (f_ / 1e9), flush=True)
run("B parent opened + closed the database")
run("B2 again, right behind B")
