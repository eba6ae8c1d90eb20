"""What ONE rank of an N-rank cooperative accelerator build does on its device, timed on a one-GPU box: the handle of rank `part` of
`n_parts` is opened on the bench's database with an exchange that moves nothing (the other ranks' regions stay unwritten, so the handle
is not searched) -- the rank's own share of the build (histogram, counts, its slices: scan, sort, fold, records), the offset lines and
the move of its region are real; the two exchanges (Lens: 4 GB, records: (N-1)/N of 4 B x entries into every rank over xGMI) are not
measurable here.      python tools/coop_part_time.py <db.edx> <K> <n_parts> [part ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BHIP_DEBUG", "1")
from burst_amd import host  # noqa: E402

edx, K, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
parts = [int(x) for x in sys.argv[4:]] or [0]
db = host.Db.read(edx)
moved = []


def nothing(ctx, base, off, part, n_parts, status):
    moved.append((off[part + 1] - off[part], off[n_parts]))
    return 1 if status else 0


cb = host.SHARE_FN(nothing)
for p in parts:
    del moved[:]
    t0 = time.time()
    dev = db.open_device_shared(0, 1, K, p, n, cb)
    dt = time.time() - t0
    print("rank %d of %d: %.2f s upload + own share of the build; own regions %s of %s bytes (Lens, records)" % (p, n, dt, [m[0] for m in moved], [m[1] for m in moved]), flush=True)
    dev.close()
