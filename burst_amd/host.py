"""ctypes binding of libburst_host.so (C host: burst_amd/csrc/host/burst_host.h) for the tests, the bench and the
multi-GPU driver.  All logic lives in C; this module only moves pointers."""
import ctypes as C
import os
import sys

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
# BURST_AMD_LIBDIR: a directory with another build of the two libraries (tools/build_prof.sh: the phase-timer build)
LIB_PATH = os.path.join(os.environ.get("BURST_AMD_LIBDIR") or _HERE, "libburst_host.so")
MODES = {"FORAGE": 0, "BEST": 1, "ALLPATHS": 2, "CAPITALIST": 3, "ANY": 4}
REP_MERGED_LIST, REP_NO_DUPE_HUNT = 1, 2

u8p, u16p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


class BhDb(C.Structure):
    _fields_ = [("rebase", C.c_int), ("xalpha", C.c_int),
                ("shear", C.c_uint32), ("totR", C.c_uint32), ("origTotR", C.c_uint32), ("numRclumps", C.c_uint32),
                ("maxLenR", C.c_uint32), ("numRefHeads", C.c_uint32),
                ("headDump", C.c_void_p), ("refHead", C.c_void_p), ("refMap", u32p), ("refStart", u32p), ("refDedupIx", u32p),
                ("tmpRIX", u32p), ("refIxSrt", u32p), ("clumpLen", u32p), ("packed", u8p), ("packedWords", C.c_uint64),
                ("hasAcx", C.c_int), ("K", C.c_int), ("acxFmt", C.c_int), ("acxZ", C.c_int),
                ("acxLens", u32p), ("acxLists", u8p), ("acxListBytes", C.c_uint64), ("badList", u32p), ("badSz", C.c_uint32),
                ("identityMap", C.c_int), ("owned", C.c_void_p * 32), ("nOwned", C.c_int), ("mapBase", C.c_void_p), ("mapLen", C.c_uint64)]


class BhQueries(C.Structure):
    _fields_ = [("totQ", C.c_uint64), ("numUniq", C.c_uint64), ("numEntries", C.c_uint64),
                ("dump", C.c_void_p), ("heads", C.c_void_p), ("offset", u64p), ("codes", u8p), ("codes4", u8p), ("codes2", u8p), ("len16", u16p), ("ambBefore", u32p), ("qoff", u64p),
                ("six", u32p), ("rc", u8p), ("flags", u8p), ("emac", u16p), ("len", u32p), ("ed", u16p),
                ("maxLen", C.c_uint32), ("minLen", C.c_uint32), ("maxED", C.c_uint32),
                ("nClear", C.c_uint64), ("nAmbig", C.c_uint64), ("nBad", C.c_uint64), ("pinned", C.c_int)]


class BhRun(C.Structure):
    _fields_ = [("hits", C.c_void_p), ("nHits", C.c_uint64), ("secAlign", C.c_double), ("total", capi.BhipStats), ("nBatches", C.c_uint32), ("hitsPinned", C.c_int), ("capHits", C.c_uint64),
                ("onBatch", C.c_void_p), ("onBatchCtx", C.c_void_p)]


ALIGN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(BhRun))
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, u8p, C.c_uint64)


class BhMultiRank(C.Structure):
    _fields_ = [("rank", C.c_int), ("hh", C.c_void_p), ("r0", u64p), ("r1", u64p), ("n_ranges", C.c_uint32), ("c0", C.c_uint32), ("run", BhRun), ("secSearch", C.c_double), ("gatherPath", C.c_int),
                ("align", ALIGN_FN), ("reduce_min", REDUCE_FN), ("ctx", C.c_void_p)]


class BhRunView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("n_runs", C.c_int), ("off", C.c_uint64 * 16), ("n", C.c_uint64 * 16), ("total", C.c_uint64)]

    def runs(self):
        """the records, run by run (numpy views of the memory they lie in)"""
        out = []
        for r in range(self.n_runs):
            n = int(self.n[r])
            if n:
                buf = (C.c_uint8 * (n * capi.HIT_DTYPE.itemsize)).from_address(self.base + int(self.off[r]) * capi.HIT_DTYPE.itemsize)
                out.append(np.frombuffer(buf, dtype=capi.HIT_DTYPE))
            else:
                out.append(np.zeros(0, capi.HIT_DTYPE))
        return out


class HostError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        capi.lib()   # libburst_hip.so first (rpath covers it, this gives the clearer error)
        L = C.CDLL(LIB_PATH)
        L.bh_last_error.restype = C.c_char_p
        L.bh_queries_sort_device.argtypes = [C.c_int]
        L.bh_queries_sort_device.restype = None
        L.bh_queries_load.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BhQueries)]
        L.bh_queries_free.argtypes = [C.POINTER(BhQueries)]
        L.bh_queries_pin.argtypes = [C.POINTER(BhQueries)]
        L.bh_edx_read.argtypes = [C.c_char_p, C.POINTER(BhDb)]
        L.bh_acx_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(BhDb)]
        L.bh_db_from_fasta.argtypes = [C.c_char_p, C.c_uint32, C.c_float, C.c_int, C.c_long, C.c_int, C.POINTER(BhDb)]
        L.bh_edx_write.argtypes = [C.POINTER(BhDb), C.c_char_p, C.c_long, C.c_float]
        L.bh_acx_build.argtypes = [C.POINTER(BhDb), C.c_int, C.c_int]
        L.bh_acx_write.argtypes = [C.POINTER(BhDb), C.c_char_p]
        L.bh_db_slice.argtypes = [C.POINTER(BhDb), C.c_uint32, C.c_uint32, C.POINTER(BhDb)]
        L.bh_db_free.argtypes = [C.POINTER(BhDb)]
        L.bh_device_open.argtypes = [C.POINTER(BhDb), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bh_device_open_ex.argtypes = [C.POINTER(BhDb), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bh_device_open_shared.argtypes = [C.POINTER(BhDb), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.bh_acx_from_device.argtypes = [C.POINTER(BhDb), C.c_void_p, C.c_int, C.c_int]
        L.bh_align.argtypes = [C.c_void_p, C.POINTER(BhQueries), C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(BhRun)]
        L.bh_align_ranges.argtypes = [C.c_void_p, C.POINTER(BhQueries), u64p, u64p, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(BhRun)]
        L.bh_align_ranges_reuse.argtypes = L.bh_align_ranges.argtypes
        L.bh_search_multi.argtypes = [C.POINTER(BhMultiRank), C.c_int, C.c_int, C.c_void_p, C.POINTER(BhQueries), C.c_int, C.c_uint64, C.c_int, C.POINTER(BhRun), u64p]
        L.bh_search_multi_ex.argtypes = [C.POINTER(BhMultiRank), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(BhQueries), C.c_int, C.c_uint64, C.c_int, C.POINTER(BhRun), u64p, C.POINTER(BhRunView)]
        L.bh_node_collect_view.argtypes = [C.c_void_p, C.POINTER(BhRunView), u64p]
        L.bh_report_view.argtypes = [C.c_void_p, C.POINTER(BhDb), C.POINTER(BhQueries), C.POINTER(BhRunView), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        L.bh_node_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
        L.bh_node_close.argtypes = [C.c_void_p]
        L.bh_node_close.restype = None
        L.bh_node_attach.argtypes = [C.c_void_p, C.POINTER(BhRun)]
        L.bh_node_attach.restype = None
        L.bh_node_begin.argtypes = [C.c_void_p]
        L.bh_node_publish.argtypes = [C.c_void_p, C.POINTER(BhRun), C.c_int]
        L.bh_node_collect.argtypes = [C.c_void_p, C.POINTER(BhRun), u64p]
        L.bh_clump_shard.argtypes = [C.POINTER(BhDb), C.c_int, C.c_int, u32p, u32p]
        L.bh_clump_shard.restype = None
        L.bh_run_reserve.argtypes = [C.POINTER(BhRun), C.c_uint64]
        L.bh_run_put.argtypes = [C.POINTER(BhRun), C.c_void_p, C.c_uint64]
        L.bh_run_reserve_plain.argtypes = [C.POINTER(BhRun), C.c_uint64]
        L.bh_run_free.argtypes = [C.POINTER(BhRun)]
        L.bh_report_ex.argtypes = [C.c_void_p, C.POINTER(BhDb), C.POINTER(BhQueries), C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.bh_synth_refs.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint64]
        L.bh_synth_reads.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32, u32p, C.c_uint32, C.c_int, C.c_double, C.c_uint64]
        L.bh_synth_refs_range.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_uint64]
        L.bh_synth_reads_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32, u32p, C.c_uint32, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_int]
        L.bh_edx_merge.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p]
        L.bh_score_lut.argtypes = [C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise HostError("libburst_host error %d: %s" % (rc, lib().bh_last_error().decode("utf-8", "replace")))


def _view(ptr, n, dtype):
    if not n or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).view(dtype)


class Db:
    def __init__(self):
        self.c = BhDb()
        self._open = False

    @classmethod
    def read(cls, edx, acx=None, K=0, z=1):
        """K = 0: the accelerator's own K (12 or 15, from its exact size)"""
        d = cls()
        _chk(lib().bh_edx_read(edx.encode(), C.byref(d.c)))
        d._open = True
        if acx:
            _chk(lib().bh_acx_read(acx.encode(), K, z, C.byref(d.c)))
        return d

    @classmethod
    def from_fasta(cls, fasta, max_len_q, thres, shear_len=500, dedupe=True, K=None, z=1):
        d = cls()
        _chk(lib().bh_db_from_fasta(fasta.encode(), max_len_q, thres, 1 if shear_len else 0, shear_len or 0, int(dedupe), C.byref(d.c)))
        d._open = True
        d.c.identityMap = 0            # a database, not a direct-FASTA run
        if K:
            _chk(lib().bh_acx_build(C.byref(d.c), K, z))
        return d

    def write(self, edx, acx=None, db_qlen=0, thres=0.97):
        _chk(lib().bh_edx_write(C.byref(self.c), edx.encode(), db_qlen, thres))
        if acx:
            _chk(lib().bh_acx_write(C.byref(self.c), acx.encode()))

    def slice(self, c0, c1):
        """view of the clumps [c0, c1) with the accelerator restricted to them (bh_db_slice); reference index of a hit
        against the slice + 16 * c0 = index in this database"""
        d = Db()
        _chk(lib().bh_db_slice(C.byref(self.c), c0, c1, C.byref(d.c)))
        d._open = True
        d._parent = self               # the view points into the parent's clump area
        return d

    def open_device(self, device=0, z=1, build_K=0):
        """build_K > 0 for a database read without its .acx: the device builds the accelerator (bhip_init with K and no tables)"""
        h = C.c_void_p()
        _chk(lib().bh_device_open_ex(C.byref(self.c), device, z, build_K, C.byref(h)))
        dev = capi.Device.__new__(capi.Device)
        dev._h = h
        dev.n_clumps = self.c.numRclumps
        dev.clump_len = _view(self.c.clumpLen, self.c.numRclumps, np.uint32)
        return dev

    def _wrap(self, h):
        dev = capi.Device.__new__(capi.Device)
        dev._h = h
        dev.n_clumps = self.c.numRclumps
        dev.clump_len = _view(self.c.clumpLen, self.c.numRclumps, np.uint32)
        return dev

    def open_devices_team(self, devices, z=1, build_K=12):
        """This (replicated) database on len(devices) handles whose accelerator the ranks build TOGETHER: one thread per rank, as
        burst_hip --gpus N does (bh_device_open_shared with bhip_team_share: every rank the lists of its share of the words, the tables
        completed device to device).  The devices may repeat (ranks sharing a device: what a one-GPU machine can test)."""
        import threading
        n = len(devices)
        team = C.c_void_p()
        capi._chk(capi.lib().bhip_team_create(n, C.byref(team)))
        hs, rcs, errs = [C.c_void_p() for _ in range(n)], [0] * n, [""] * n
        share = C.cast(capi.lib().bhip_team_share, C.c_void_p)
        def work(r):
            rcs[r] = lib().bh_device_open_shared(C.byref(self.c), devices[r], z, build_K, r, n, share, team, C.byref(hs[r]))
            if rcs[r]:
                errs[r] = lib().bh_last_error().decode()
        ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        capi.lib().bhip_team_destroy(team)
        devs = [self._wrap(h) if not rc else None for h, rc in zip(hs, rcs)]
        if any(rcs):
            for d in devs:
                if d is not None:
                    d.close()
            raise HostError(next(rc for rc in rcs if rc), "; ".join(e for e in errs if e))
        return devs

    def open_device_shared(self, device, z, build_K, part, n_parts, share, ctx=None):
        """one of n_parts handles (one process each) that build the accelerator together; `share` = a bhip_share_fn (dist_share(...),
        or capi.lib().bhip_comm_share with ctx = a BhipCommRank)"""
        h = C.c_void_p()
        _chk(lib().bh_device_open_shared(C.byref(self.c), device, z, build_K, part, n_parts, C.cast(share, C.c_void_p), ctx, C.byref(h)))
        return self._wrap(h)

    def acx_from_device(self, dev, K, z=1):
        """accelerator tables of this database from a device handle that built them (bh_acx_from_device): write() then saves the .acx"""
        _chk(lib().bh_acx_from_device(C.byref(self.c), dev._h, K, z))

    def acx_write_from_device(self, dev, K, path, z=1):
        """the .acx of this database written straight from the tables a device handle built, list area streamed (bh_acx_write_from_device)"""
        _chk(lib().bh_acx_write_from_device(C.byref(self.c), dev._h, K, z, path.encode()))

    def close(self):
        if self._open:
            lib().bh_db_free(C.byref(self.c))
            self._open = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuerySet:
    def __init__(self, fasta, thres, rc=False, accel=True, K=12, z=1, whitespace=False):
        self.c = BhQueries()
        _chk(lib().bh_queries_load(fasta.encode(), thres, int(rc), int(whitespace), int(accel), K, z, 0, C.byref(self.c)))
        c = self.c
        self.n_reads, self.n_uniq, self.n_entries = int(c.totQ), int(c.numUniq), int(c.numEntries)

    def batch(self, u0=0, u1=None):
        """capi.Queries view (no copy for the forward-only case) of unique queries [u0, u1) with their RC twins"""
        c = self.c
        u1 = self.n_uniq if u1 is None else min(u1, self.n_uniq)
        qoff = _view(c.qoff, self.n_entries + 1, np.uint64)
        codes = _view(c.codes, int(qoff[-1]), np.uint8)
        ent = np.arange(u0, u1, dtype=np.int64)
        if self.n_entries > self.n_uniq:
            ent = np.concatenate([ent, ent + self.n_uniq])
        q = capi.Queries.__new__(capi.Queries)
        lens = (qoff[ent + 1] - qoff[ent]).astype(np.uint64)
        q.off = np.zeros(len(ent) + 1, np.uint64)
        np.cumsum(lens, out=q.off[1:])
        if self.n_entries == self.n_uniq:
            q.codes = codes[int(qoff[u0]):int(qoff[u1])]
        else:
            q.codes = np.concatenate([codes[int(qoff[u0]):int(qoff[u1])], codes[int(qoff[self.n_uniq + u0]):int(qoff[self.n_uniq + u1])]])
        q.emac = np.ascontiguousarray(_view(c.emac, self.n_entries, np.uint16)[ent])
        q.six = (np.ascontiguousarray(_view(c.six, self.n_entries, np.uint32)[ent]) - np.uint32(u0)).astype(np.uint32)
        q.rc = np.ascontiguousarray(_view(c.rc, self.n_entries, np.uint8)[ent])
        q.flags = np.ascontiguousarray(_view(c.flags, self.n_entries, np.uint8)[ent])
        q.n = len(ent)
        q.n_shared = u1 - u0
        q.entry_index = ent
        return q

    def pin(self):
        """page-lock the arrays the batches are copied from (bh_queries_pin)"""
        _chk(lib().bh_queries_pin(C.byref(self.c)))

    def reads_in(self, u0, u1):
        off = _view(self.c.offset, self.n_uniq + 1, np.uint64)
        return int(off[min(u1, self.n_uniq)] - off[u0])

    def close(self):
        if self.c.codes:
            lib().bh_queries_free(C.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Run:
    """records and statistics of bh_align / bh_align_ranges (the C batch scheduler: double-buffered staging, asynchronous
    hand-over of the records)"""

    def __init__(self):
        self.c = BhRun()

    @property
    def hits(self):
        n = int(self.c.nHits)
        if not n:
            return np.zeros(0, capi.HIT_DTYPE)
        buf = (C.c_uint8 * (n * capi.HIT_DTYPE.itemsize)).from_address(self.c.hits)
        return np.frombuffer(buf, dtype=capi.HIT_DTYPE)

    def stats(self):
        return self.c.total.as_dict()

    def reserve(self, cap_records):
        _chk(lib().bh_run_reserve(C.byref(self.c), int(cap_records)))

    def close(self):
        lib().bh_run_free(C.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_ranges(dev, qs, ranges, mode, batch_uniq=1 << 18, run=None):
    """bh_align_ranges: ranges = [(u0, u1), ...] of unique queries, aligned in order through the device handle of `dev`;
    run = a Run used before: its page-locked record buffer is reused (bh_align_ranges_reuse)"""
    r0 = np.ascontiguousarray([r[0] for r in ranges], np.uint64)
    r1 = np.ascontiguousarray([r[1] for r in ranges], np.uint64)
    run = run or Run()
    _chk(lib().bh_align_ranges_reuse(dev._h, C.byref(qs.c), r0.ctypes.data_as(u64p), r1.ctypes.data_as(u64p), len(ranges), MODES[mode], batch_uniq, C.byref(run.c)))
    return run


class Node:
    """the shared-memory hand-over of the records between the ranks of one node that live in different processes (bh_node.c):
    every rank's record buffer is a segment rank 0 maps; no collective"""

    def __init__(self, job, rank, world, cap_records):
        self.h = C.c_void_p()
        _chk(lib().bh_node_open(job.encode(), rank, world, int(cap_records), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().bh_node_close(self.h)
            self.h = C.c_void_p()


class RankSearch:
    """this process's rank of a node-wide job through the C host's multi-GPU search (bh_search_multi_ex: align, [database-sharded:
    per-query minimum over the ranks,] the records to rank 0 -- through the shared-memory segments of `node`, or gathered over the
    library's RCCL communicator)"""

    def __init__(self, dev, rank, world, comm, c0=0, node=None, align=None, reduce_min=None):
        """align(ranges, mode_number) -> HIT_DTYPE records (q = global entry index, sorted by (q, refIx)): a back end in place of the
        device scheduler (the CPU tests put the oracle here; dev may then be None).  reduce_min(uint8 array) -> None: the element-wise
        minimum over all ranks, in place (database-sharded ranks in different processes without a library communicator: the
        launcher's own collective)"""
        self.mr = BhMultiRank()
        self.mr.rank, self.mr.hh, self.mr.c0 = rank, (dev._h if dev is not None else None), c0
        self.world, self.comm, self.node = world, comm, node
        self.all = Run()
        self._cb = []
        if align is not None:
            def _align(ctx, q, u0, u1, n, mode, batch, run):
                try:
                    h = np.ascontiguousarray(align([(int(u0[i]), int(u1[i])) for i in range(n)], int(mode)), dtype=capi.HIT_DTYPE)
                    return int(lib().bh_run_put(run, h.ctypes.data_as(C.c_void_p), len(h)))
                except Exception as e:      # (an exception must not cross the C frame)
                    import traceback
                    traceback.print_exc()
                    return -5
            self._cb.append(ALIGN_FN(_align))
            self.mr.align = self._cb[-1]
        if reduce_min is not None:
            def _reduce(ctx, buf, n):
                try:
                    a = np.ctypeslib.as_array(buf, shape=(int(n),))
                    reduce_min(a)
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return -5
            self._cb.append(REDUCE_FN(_reduce))
            self.mr.reduce_min = self._cb[-1]

    def reserve(self, cap_records):
        if self.node is None:      # (with a node the rank's buffer is its segment, sized at bh_node_open)
            _chk(lib().bh_run_reserve(C.byref(self.mr.run), int(cap_records)))

    def reserve_all(self, cap_records, pinned=False):
        """rank 0: room for everybody's records (page-locked when a device copy lands in it: the RCCL gather)"""
        _chk((lib().bh_run_reserve if pinned else lib().bh_run_reserve_plain)(C.byref(self.all.c), int(cap_records)))

    def search(self, qs, ranges, mode, batch_uniq, shard_db=False):
        """ranges: this rank's [(u0, u1), ...]; returns the gathered Run on rank 0 (its own elsewhere)"""
        r0 = np.ascontiguousarray([r[0] for r in ranges] or [0], np.uint64)
        r1 = np.ascontiguousarray([r[1] for r in ranges] or [0], np.uint64)
        self.mr.r0, self.mr.r1, self.mr.n_ranges = r0.ctypes.data_as(u64p), r1.ctypes.data_as(u64p), len(ranges)
        self._keep = (r0, r1)
        self.counts = np.zeros(self.world, np.uint64)
        self.view = BhRunView()
        n_shards = int(shard_db) if shard_db is not True else self.world      # (True: every rank its own shard)
        _chk(lib().bh_search_multi_ex(C.byref(self.mr), 1, self.world, self.comm, self.node.h if self.node is not None else None, C.byref(qs.c), MODES[mode], batch_uniq,
                                      n_shards, C.byref(self.all.c), self.counts.ctypes.data_as(u64p), C.byref(self.view)))
        return self.all      # (rank 0: self.view says where the records are -- with a node they stay in the ranks' segments)

    def own_stats(self):
        return self.mr.run.total.as_dict(), int(self.mr.run.nBatches), float(self.mr.run.secAlign)

    def close(self):
        lib().bh_run_free(C.byref(self.mr.run))
        self.all.close()
        if self.node is not None:
            self.node.close()


libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]


SHARE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int)


def share_regions(off, part, n_parts, status, fetch, store, broadcast, any_failed, piece=64 << 20):
    """The exchange of the cooperative accelerator build (bhip_share_fn, include/burst_hip.h) over a launcher's own collectives, for ranks
    that have no RCCL communicator (the gloo back end; ranks sharing a device): region r of every rank's array, [off[r], off[r + 1]),
    travels from its builder to everybody, piece by piece through host memory.  fetch(a, n) -> bytes of the own array as a uint8 array,
    store(a, buf), broadcast(buf, root) -> buf, any_failed(status) -> bool.  Returns 0, 1 when some rank announced a failure before the
    exchange, -1 (on every rank) when a rank failed inside it."""
    if any_failed(status):
        return 1
    # A local failure in the middle (a device copy) must not leave the peers waiting in the next broadcast: the rank notes it, keeps
    # taking part with whatever buffer it has, and the ranks agree on the outcome in one more reduction at the end.
    broken = None
    for r in range(n_parts):
        a = int(off[r])
        while a < int(off[r + 1]):
            n = min(piece, int(off[r + 1]) - a)
            buf = np.empty(n, np.uint8)
            if r == part and broken is None:
                try:
                    buf = fetch(a, n)
                except Exception as e:
                    broken = e
            buf = broadcast(buf, r)
            if r != part and broken is None:
                try:
                    store(a, buf)
                except Exception as e:
                    broken = e
            a += n
    if broken is not None:
        sys.stderr.write("share_regions: rank %d: %s\n" % (part, broken))
    return -1 if any_failed(broken is not None) else 0


def dist_share(dist, pdev="cpu"):
    """bhip_share_fn over torch.distributed (any back end): returns the ctypes callback object for Db.open_device_shared -- the CALLER keeps
    it alive (a reference held) for as long as the library may call it, i.e. across bh_device_open_shared"""
    import torch
    def any_failed(status):
        t = torch.tensor([1 if status else 0], dtype=torch.int64, device=pdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))
    def broadcast(buf, root):
        t = torch.from_numpy(buf).to(pdev)
        dist.broadcast(t, root)
        return t.cpu().numpy()
    class _Raw:          # a piece of the device array as a tensor, no copy (the nccl back end broadcasts it in place: RCCL over xGMI)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    def cb(ctx, base, off, part, n_parts, status):
        try:
            if pdev != "cpu":
                if any_failed(status):
                    return 1
                for r in range(n_parts):
                    a = int(off[r])
                    while a < int(off[r + 1]):
                        n = min(1 << 30, int(off[r + 1]) - a)
                        dist.broadcast(torch.as_tensor(_Raw(base + a, n), device=pdev), r)
                        a += n
                torch.cuda.synchronize()
                return 0
            def fetch(a, n):
                out = np.empty(n, np.uint8)
                capi._chk(capi.lib().bhip_device_copy(out.ctypes.data, base + a, n, 0))
                return out
            def store(a, buf):
                buf = np.ascontiguousarray(buf)
                capi._chk(capi.lib().bhip_device_copy(base + a, buf.ctypes.data, len(buf), 1))
            return share_regions([off[i] for i in range(n_parts + 1)], part, n_parts, status, fetch, store, broadcast, any_failed)
        except Exception as e:          # (an exception must not unwind through the C caller)
            sys.stderr.write("dist_share: %s\n" % e)
            return -1
    return SHARE_FN(cb)


def shard_range(n_uniq, world, rank):
    """unique queries of rank `rank` of `world`: contiguous, sizes differing by at most one (a query and its reverse-complement twin
    are one unique query: they stay together)"""
    base, rem = divmod(n_uniq, world)
    u0 = rank * base + min(rank, rem)
    return u0, u0 + base + (1 if rank < rem else 0)


def clump_shard(db, world, rank):
    """clump range of database shard `rank` of `world` (bh_clump_shard: about the same number of reference columns each)"""
    c0, c1 = C.c_uint32(), C.c_uint32()
    lib().bh_clump_shard(C.byref(db.c), world, rank, C.byref(c0), C.byref(c1))
    return int(c0.value), int(c1.value)


def report(path, db, qs, hits, mode, flags=0):
    """hits: HIT_DTYPE array with q = global entry index, records of one entry contiguous"""
    f = libc.fopen(path.encode(), b"wb")
    if not f:
        raise HostError("cannot open %s" % path)
    n = C.c_uint64()
    hits = np.ascontiguousarray(hits)
    try:
        _chk(lib().bh_report_ex(f, C.byref(db.c), C.byref(qs.c), hits.ctypes.data_as(C.c_void_p), len(hits), MODES[mode], flags, C.byref(n)))
    finally:
        libc.fclose(f)
    return int(n.value)


def report_view(path, db, qs, view, mode, flags=0):
    """the same from a BhRunView (records in several runs: RankSearch.view)"""
    f = libc.fopen(path.encode(), b"wb")
    if not f:
        raise HostError("cannot open %s" % path)
    n = C.c_uint64()
    try:
        _chk(lib().bh_report_view(f, C.byref(db.c), C.byref(qs.c), C.byref(view), MODES[mode], flags, None, C.byref(n)))
    finally:
        libc.fclose(f)
    return int(n.value)


def synth_refs(path, n_base, n_variants, length, rate, seed, first_base=0):
    """base sequences [first_base, first_base + n_base) with their variants: a sequence depends on the seed and its number only"""
    _chk(lib().bh_synth_refs_range(path.encode(), first_base, n_base, n_variants, length, rate, seed))


def edx_merge(paths, out):
    """several .edx files laid end to end into one (bh_edx_merge)"""
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    _chk(lib().bh_edx_merge(arr, len(paths), out.encode()))


def synth_reads(refs, path, n_reads, read_len, edits, rc=False, iupac=0.0, seed=42, first_read=0, append=False):
    e = np.ascontiguousarray(edits, np.uint32)
    _chk(lib().bh_synth_reads_ex(refs.encode(), path.encode(), n_reads, read_len, e.ctypes.data_as(u32p), len(e), int(rc), iupac, seed, first_read, int(append)))
