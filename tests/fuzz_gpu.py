"""Randomised parity run on the GPU box: random small databases / read sets / options, device records vs the oracle.
   python tests/fuzz_gpu.py [seconds] [seed]   -- exits non-zero and prints the failing configuration on the first mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_kernels as T          # family_db, make_queries, oracle_hits
import dbutil, oraclelib as ol
from burst_amd import capi
capi.LOAD_LEGACY_PREFILTERS = True          # (the fuzzer also draws the superseded prefilter kernels: the test-only library in front of the product's)

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
t0 = time.time(); it = 0; recs = 0
while time.time() - t0 < budget_s:
    it += 1
    cfg = dict(seed=int(rng.integers(1 << 30)), n_base=int(rng.integers(1, 8)), n_var=int(rng.choice([3, 9, 17, 40, 120])), length=int(rng.integers(150, 700)),
               rate=float(rng.choice([0.01, 0.03, 0.06, 0.1])), short=bool(rng.integers(2)), db_iupac=float(rng.choice([0, 0, 0.003])),
               qlen=int(rng.choice([20, 33, 64, 65, 100, 100, 128, 150, 200, 250, 292, 320, 400])), thres=float(rng.choice([0.9, 0.93, 0.95, 0.97, 0.98, 0.99])),
               q_iupac=float(rng.choice([0, 0, 0.01, 0.03])), K=int(rng.choice([8, 10, 12, 12])), fmt=int(rng.integers(2)), accel=bool(rng.integers(4) > 0),
               all_hits=bool(rng.integers(2)), nq=int(rng.integers(5, 60)),
               stride=int(rng.choice([0, 0, 0, 1, 3, 7, 12, 18])), algo=int(rng.integers(-1, 2)), table=int(rng.choice([0, 0, 9, 10, 11])), reg=int(rng.integers(2)), lanes=int(rng.choice([1, 1, 2, 5])),
               prune=int(rng.integers(3) > 0), two_stage=int(rng.integers(4) > 0), lane_masks=int(rng.integers(4) > 0), y=int(rng.integers(4) == 0), band=int(rng.integers(4) > 0), oversub=int(rng.choice([1, 2, 2, 5])), cw=int(rng.integers(3)))
    if os.environ.get("FUZZ_LOG"):                  # (a crash of the device leaves no Python frame behind: the configuration in hand, line by line)
        with open(os.environ["FUZZ_LOG"], "a") as lf: lf.write("%d %r\n" % (it, cfg))
    seqs = T.family_db(cfg["seed"], cfg["n_base"], cfg["n_var"], cfg["length"], rate=cfg["rate"], short=cfg["short"], iupac=cfg["db_iupac"])
    if cfg["qlen"] + 10 > min(len(s) for s in seqs):
        cfg["qlen"] = max(20, min(len(s) for s in seqs) - 10)
    packed, clump_len, tot = dbutil.pack_clumps(seqs)
    lut = ol.score_lut(0 if cfg["y"] else 1)
    kw = {}
    if cfg["accel"]:
        # clumps holding an ambiguous reference symbol go to the BadList (aligned unconditionally, burst.c:3432-3437): the test
        # builder does not expand ambiguous reference words the way make_accelerator does, so they must not rely on votes
        bad = sorted({i // 16 for i, s in enumerate(seqs) if (np.asarray(s) > 4).any()})
        lens, entries, offs = dbutil.build_acx(seqs, cfg["K"], skip_clumps=tuple(bad))
        kw = dict(acx_lens=lens, acx_lists=dbutil.pack_acx_lists(lens, entries, cfg["fmt"]), acx_fmt=cfg["fmt"], K=cfg["K"],
                  badlist=np.array(bad, np.uint32) if bad else None)
    lens = [cfg["qlen"]]
    if rng.integers(3) == 0:                       # a batch of mixed lengths: several length classes in one call
        lens += [int(x) for x in rng.choice([24, 40, 70, 100, 130, 200, 260], size=int(rng.integers(1, 3)))]
    lens = [min(L, max(20, min(len(s) for s in seqs) - 10)) for L in lens]
    cfg["lens"] = lens
    from burst_amd import synth as _synth
    reads = []
    for li_, L in enumerate(lens):
        bud = T.budget(cfg["thres"], L)
        r, _ = _synth.make_reads(seqs, max(2, cfg["nq"] // len(lens)), L, [0, 1, 2, max(0, bud), bud + 2], cfg["seed"] + 1 + li_, rc_frac=0.5, iupac_frac=cfg["q_iupac"])
        reads += r
    cfg["junk"] = int(rng.integers(4) == 0)
    if cfg["junk"]:                                # symbols of code 0 (outside the alphabet: cost 255, can only face a gap) inside some reads;
        for ri_ in range(0, len(reads), 3):        # never first or last, where the reference's re-scorer stops (burst.c:812-816)
            r_ = np.array(reads[ri_], np.uint8)
            if len(r_) > 8:
                r_[1 + rng.choice(len(r_) - 2, size=int(rng.integers(1, 3)), replace=False)] = 0
            reads[ri_] = r_
    nq_ = len(reads)
    allq = reads + [_synth.revcomp(r) for r in reads]
    q = capi.Queries(allq, [T.budget(cfg["thres"], len(r)) for r in reads] * 2, list(range(nq_)) * 2, [0] * nq_ + [1] * nq_)
    if cfg["accel"]:
        q.flags = np.zeros(q.n, np.uint8)
    for kv in filter(None, os.environ.get("FUZZ_OVERRIDE", "").split(",")):      # e.g. FUZZ_OVERRIDE=cw=2,lanes=1 while narrowing a failure down
        cfg[kv.split("=")[0]] = type(cfg[kv.split("=")[0]])(int(kv.split("=")[1]))
    if os.environ.get("FUZZ_LOG"):
        with open(os.environ["FUZZ_LOG"], "a") as lf: lf.write("%d FULL %r\n" % (it, cfg))
    if it < int(os.environ.get("FUZZ_SKIP_TO", "0")):      # fast-forward to a configuration of a logged run: the same draws, no device work
        if not cfg["all_hits"] and rng.integers(2):
            rng.permutation(tot)
        continue
    dev = capi.Device(packed, clump_len, tot, lut, **kw)
    try:
        for name, key in (("prefilter_stride", "stride"), ("prefilter_algo", "algo"), ("prefilter_table", "table"), ("rescore_reg", "reg"), ("lanes", "lanes"), ("two_stage", "two_stage"), ("lane_masks", "lane_masks"), ("prune", "prune"), ("band", "band"), ("oversub", "oversub"), ("prefilter_cw", "cw")):
            dev.set_option(name, cfg[key])
        dev.set_option("lane_min_entries", 4)
        got = dev.align_batch(q, all_hits=cfg["all_hits"])
        exp = T.oracle_hits(packed, clump_len, tot, q, lut, cfg["all_hits"])
        if len(got) != len(exp) or got.tobytes() != exp.tobytes():
            print("MISMATCH at iteration %d: %r\n got %d records, expected %d" % (it, cfg, len(got), len(exp)))
            n = min(len(got), len(exp))
            bad = [i for i in range(n) if got[i].tobytes() != exp[i].tobytes()][:5]
            for i in bad:
                print("  first differing record", i, got[i], exp[i])
            sys.exit(1)
        recs += len(exp)
        if not cfg["all_hits"] and rng.integers(2):
            # BEST chosen on the device (BHIP_HITS_BEST) under the same random options: the reference's choice computed from the oracle's records
            order = rng.permutation(tot).astype(np.uint32)
            dev.set_ref_order(order)
            expb = T.best_per_entry(exp, order)
            gotb = dev.align_batch(q, all_hits=2)
            if len(gotb) != len(expb) or gotb.tobytes() != expb.tobytes():
                print("MISMATCH (BEST on the device) at iteration %d: %r\n got %d records, expected %d" % (it, cfg, len(gotb), len(expb)))
                sys.exit(1)
            recs += len(expb)
    finally:
        dev.close()
    if it >= int(os.environ.get("FUZZ_STOP_AFTER", "1000000000")):
        break
print("fuzz ok: %d configurations, %d records compared in %.0f s (seed %d)" % (it, recs, time.time() - t0, seed0))
