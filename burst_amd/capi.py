"""ctypes binding of libburst_hip.so (C ABI declared in include/burst_hip.h).

This is the only door from Python to the device path; there is no CPU fallback.  If the library has not
been built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C burst_amd/csrc`) importing
this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BURST_AMD_LIBDIR: a directory with another build of the two libraries (tools/build_prof.sh: the phase-timer build)
LIB_PATH = os.path.join(os.environ.get("BURST_AMD_LIBDIR") or _HERE, "libburst_hip.so")

BHIP_OK, BHIP_E_ARG, BHIP_E_DEVICE, BHIP_E_CAPACITY, BHIP_E_QUERYLEN, BHIP_E_INTERNAL, BHIP_E_RESCORE = 0, -1, -2, -3, -4, -5, -6
BHIP_Q_PREFILTER, BHIP_Q_EXHAUSTIVE = 0, 1
BHIP_MAX_QLEN = 4095

# BhipHit, 20 bytes (include/burst_hip.h)
HIT_DTYPE = np.dtype([("q", "<u4"), ("refIx", "<u4"), ("finalPos", "<u4"), ("score", "<f4"),
                      ("ed", "u1"), ("gapR", "u1"), ("gapQ", "u1"), ("rc", "u1")])
assert HIT_DTYPE.itemsize == 20


class BhipStats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_pairs", C.c_uint64), ("n_columns", C.c_uint64),
                ("n_raw_hits", C.c_uint64), ("n_hits", C.c_uint64), ("acx_entries_read", C.c_uint64),
                ("bytes_algorithmic", C.c_uint64), ("n_windows", C.c_uint64), ("n_window_columns", C.c_uint64),
                ("n_lane_tasks", C.c_uint64), ("n_task_columns", C.c_uint64), ("n_seed_words", C.c_uint64),
                ("ms_h2d", C.c_float), ("ms_prefilter", C.c_float), ("ms_peq", C.c_float), ("ms_myers", C.c_float),
                ("ms_rescore", C.c_float), ("ms_d2h", C.c_float), ("ms_total", C.c_float),
                ("ms_myers_prefix", C.c_float), ("ms_myers_window", C.c_float), ("ms_prefilter_hash", C.c_float), ("ms_seed", C.c_float),
                ("ms_stage_copy", C.c_float), ("ms_stage_route", C.c_float),
                ("myers_launches", C.c_uint32), ("prefix_words", C.c_uint32), ("prefilter_launches", C.c_uint32), ("prefilter_algo", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class BhipQuerySpan(C.Structure):
    _fields_ = [("codes", C.c_void_p), ("codes4", C.c_void_p), ("off", C.c_void_p), ("emac", C.c_void_p), ("rc", C.c_void_p), ("flags", C.c_void_p),
                ("n", C.c_uint32), ("q_base", C.c_uint32), ("codes2", C.c_void_p), ("len", C.c_void_p)]


EXPORTS = ["bhip_init", "bhip_stage_queries", "bhip_align_staged", "bhip_align_batch", "bhip_align_pairs", "bhip_prefilter", "bhip_set_option", "bhip_get_stats",
           "bhip_device_info", "bhip_destroy", "bhip_last_error", "bhip_abi_version", "bhip_set_ref_order", "bhip_copy_hits_device", "bhip_sync_hits",
           "bhip_comm_create", "bhip_comm_unique_id", "bhip_comm_create_rank", "bhip_comm_allreduce_min", "bhip_comm_fetch_gathered", "bhip_comm_gather_hits", "bhip_comm_stage_device", "bhip_comm_gather_staged", "bhip_comm_stage_reset", "bhip_comm_destroy", "bhip_acx_export", "bhip_reserve", "bhip_reserve_symbols", "bhip_sort_queries", "bhip_stage_spans", "bhip_alloc_host", "bhip_free_host", "bhip_host_register", "bhip_host_unregister", "bhip_set_enqueued_hook", "bhip_acx_export_entries",
           "bhip_build_accelerator_shared", "bhip_comm_share", "bhip_team_create", "bhip_team_destroy", "bhip_team_share", "bhip_device_copy"]


class BurstHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libburst_hip error %d: %s" % (code, msg))
        self.code = code


LOAD_LEGACY_PREFILTERS = False


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: the HIP extension has not been built (run __graft_entry__.build()); "
                          "burst_amd has no CPU fallback" % LIB_PATH)
    # The superseded prefilter kernels (options prefilter_cw = 0 / 1) are NOT in libburst_hip.so: the tests ask for the test-only library
    # that holds them (tests/conftest.py sets LOAD_LEGACY_PREFILTERS in THIS process; child processes -- bench.py, burst_hip -- run the
    # product library alone).  Loaded first and globally, so that the product library's two weak references (bhip_internal.h:
    # bhip_legacy_pf_launch / _attrs) resolve to it.
    legacy = os.path.join(os.path.dirname(LIB_PATH), "libburst_hip_legacy.so")
    if LOAD_LEGACY_PREFILTERS and os.path.exists(legacy):
        globals()["_legacy_lib"] = C.CDLL(legacy, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    lib.bhip_init.argtypes = [i32, vp, vp, u32, u32, vp, vp, i32, i32, vp, u32, vp, i32, C.POINTER(vp)]
    lib.bhip_init.restype = i32
    lib.bhip_align_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32, i32, vp, u64, C.POINTER(u64)]
    lib.bhip_align_batch.restype = i32
    lib.bhip_stage_queries.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u32]
    lib.bhip_stage_queries.restype = i32
    lib.bhip_align_staged.argtypes = [vp, i32, vp, u64, C.POINTER(u64)]
    lib.bhip_align_staged.restype = i32
    lib.bhip_align_pairs.argtypes = [vp, vp, vp, vp, u32, vp, vp, u64, vp]
    lib.bhip_align_pairs.restype = i32
    lib.bhip_prefilter.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, u64, C.POINTER(u64)]
    lib.bhip_prefilter.restype = i32
    lib.bhip_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    lib.bhip_set_option.restype = i32
    lib.bhip_get_stats.argtypes = [vp, C.POINTER(BhipStats)]
    lib.bhip_get_stats.restype = i32
    lib.bhip_device_info.argtypes = [vp, C.c_char_p, i32, C.POINTER(i32), C.POINTER(u64)]
    lib.bhip_device_info.restype = i32
    lib.bhip_copy_hits_device.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.bhip_copy_hits_device.restype = i32
    lib.bhip_sync_hits.argtypes = [vp]
    lib.bhip_sync_hits.restype = i32
    lib.bhip_destroy.argtypes = [vp]
    lib.bhip_destroy.restype = None
    lib.bhip_last_error.argtypes = []
    lib.bhip_last_error.restype = C.c_char_p
    lib.bhip_set_ref_order.argtypes = [vp, vp, u32]
    lib.bhip_set_ref_order.restype = i32
    lib.bhip_abi_version.argtypes = []
    lib.bhip_abi_version.restype = i32
    lib.bhip_stage_spans.argtypes = [vp, C.POINTER(BhipQuerySpan), u32, u32, u32]
    lib.bhip_reserve.argtypes = [vp, u32, u32]
    lib.bhip_reserve.restype = i32
    lib.bhip_stage_spans.restype = i32
    lib.bhip_reserve_symbols.argtypes = [vp, u32, u32, u64]
    lib.bhip_reserve_symbols.restype = i32
    lib.bhip_alloc_host.argtypes = [u64]
    lib.bhip_alloc_host.restype = vp
    lib.bhip_free_host.argtypes = [vp]
    lib.bhip_free_host.restype = None
    lib.bhip_host_register.argtypes = [vp, u64]
    lib.bhip_host_register.restype = i32
    lib.bhip_host_unregister.argtypes = [vp]
    lib.bhip_host_unregister.restype = i32
    lib.bhip_comm_unique_id.argtypes = [vp]
    lib.bhip_comm_unique_id.restype = i32
    lib.bhip_comm_create_rank.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    lib.bhip_comm_create_rank.restype = i32
    lib.bhip_comm_destroy.argtypes = [vp]
    lib.bhip_comm_destroy.restype = None
    lib.bhip_acx_export.argtypes = [vp, vp, vp, vp, u64, C.POINTER(u64), vp, u32, C.POINTER(u32)]
    lib.bhip_acx_export.restype = i32
    lib.bhip_build_accelerator_shared.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.bhip_build_accelerator_shared.restype = i32
    lib.bhip_team_create.argtypes = [i32, C.POINTER(vp)]
    lib.bhip_team_create.restype = i32
    lib.bhip_team_destroy.argtypes = [vp]
    lib.bhip_team_destroy.restype = None
    lib.bhip_device_copy.argtypes = [vp, vp, u64, i32]
    lib.bhip_device_copy.restype = i32
    return lib


_lib = None


def _hits_mode(all_hits):
    """False / True (every hit within budget) as before; 2 or "best" = BHIP_HITS_BEST (one record per entry, chosen on the device)"""
    return 2 if all_hits in (2, "best") else int(bool(all_hits))


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc):
    if rc != BHIP_OK:
        raise BurstHipError(rc, lib().bhip_last_error().decode("utf-8", "replace"))


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Queries:
    """Flat query batch in the layout bhip_align_batch takes."""

    def __init__(self, seqs, emac, six=None, rc=None, flags=None):
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        self.off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=self.off[1:])
        self.codes = (np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) and self.off[-1]
                      else np.zeros(1, np.uint8))
        self.emac = _arr(emac, np.uint16)
        self.six = _arr(six, np.uint32)
        self.rc = _arr(rc, np.uint8)
        self.flags = _arr(flags, np.uint8)
        self.n = len(seqs)
        self.n_shared = int(self.six.max()) + 1 if self.six is not None and self.n else self.n


class Device:
    """One handle = one database resident on one GPU (bhip_init .. bhip_destroy)."""

    def __init__(self, edx_packed, clump_len, tot_refs, score_lut, acx_lens=None, acx_lists=None, acx_fmt=0, K=12,
                 badlist=None, device=0, xalpha=0, build_acx=False):
        """acx_lens / acx_lists: the tables of an .acx file; build_acx=True (and no tables): the device builds the accelerator
        for word length K from the references alone"""
        self._h = C.c_void_p()
        if acx_lens is None and not build_acx:
            K = 0
        edx_packed = _arr(edx_packed, np.uint8)
        clump_len = _arr(clump_len, np.uint32)
        score_lut = _arr(score_lut, np.uint8)
        acx_lens = _arr(acx_lens, np.uint32)
        acx_lists = _arr(acx_lists, np.uint8)
        badlist = _arr(badlist, np.uint32)
        nbad = 0 if badlist is None else len(badlist)
        self.n_clumps = len(clump_len)
        self.clump_len = clump_len
        _chk(lib().bhip_init(device, _ptr(edx_packed), _ptr(clump_len), len(clump_len), tot_refs, _ptr(acx_lens), _ptr(acx_lists),
                             acx_fmt, K, _ptr(badlist), nbad, _ptr(score_lut), xalpha, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().bhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int()
        hbm = C.c_uint64()
        _chk(lib().bhip_device_info(self._h, name, 256, C.byref(ncu), C.byref(hbm)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": hbm.value}

    def acx_export(self, K, masks=True):
        """the handle's accelerator in the file's terms: (Lens[4^K], clump ids in word order, lane masks or None, BadList)"""
        n, nb = C.c_uint64(), C.c_uint32()
        _chk(lib().bhip_acx_export(self._h, None, None, None, 0, C.byref(n), None, 0, C.byref(nb)))
        lens = np.zeros(1 << (2 * K), np.uint32)
        clumps = np.zeros(max(1, n.value), np.uint32)
        mk = np.zeros(max(1, n.value), np.uint16) if masks else None
        bad = np.zeros(max(1, nb.value), np.uint32)
        _chk(lib().bhip_acx_export(self._h, _ptr(lens), _ptr(clumps), _ptr(mk), n.value, C.byref(n), _ptr(bad), nb.value, C.byref(nb)))
        return lens, clumps[:n.value], (mk[:n.value] if masks else None), bad[:nb.value]

    def set_option(self, name, value):
        _chk(lib().bhip_set_option(self._h, name.encode(), int(value)))

    def reserve(self, n_entries, max_len):
        """bhip_reserve: size the device buffers for batches of n_entries query entries of at most max_len symbols and run the
        library's warm-up pass (what the burst_hip command line does once per job, before the search phase)"""
        _chk(lib().bhip_reserve(self._h, int(n_entries), int(max_len)))

    def stats(self, raw=False):
        """statistics of the last call (raw=True: the ctypes struct, no dict -- for timed loops)"""
        s = BhipStats()
        _chk(lib().bhip_get_stats(self._h, C.byref(s)))
        return s if raw else s.as_dict()

    def set_ref_order(self, order):
        """BEST's tie-break table (order[refIx] = RefIxSrt[refIx]) for all_hits = 2 / "best" (bhip_set_ref_order)"""
        order = _arr(order, np.uint32)
        _chk(lib().bhip_set_ref_order(self._h, _ptr(order), len(order)))

    def align_batch(self, q, all_hits=False, cap=None):
        cap = cap or max(1 << 16, 4 * q.n)
        while True:
            hits = np.zeros(cap, dtype=HIT_DTYPE)
            n = C.c_uint64()
            rc = lib().bhip_align_batch(self._h, _ptr(q.codes), _ptr(q.off), _ptr(q.emac), _ptr(q.six), _ptr(q.rc), _ptr(q.flags),
                                        q.n, q.n_shared, _hits_mode(all_hits), _ptr(hits), cap, C.byref(n))
            if rc == BHIP_E_CAPACITY:
                cap = int(n.value) + 16
                continue
            _chk(rc)
            return hits[:n.value]

    def stage(self, q):
        """upload a batch once (bhip_stage_queries); align_staged() then runs on the resident copy"""
        self._staged = q     # keep the host arrays alive
        _chk(lib().bhip_stage_queries(self._h, _ptr(q.codes), _ptr(q.off), _ptr(q.emac), _ptr(q.six), _ptr(q.rc), _ptr(q.flags), q.n, q.n_shared))

    def stage_spans(self, spans, n_shared, max_len=0):
        """asynchronous staging (bhip_stage_spans).  spans: list of dicts with numpy arrays codes, off (n + 1 offsets into codes),
        emac and optional rc / flags, plus q_base; entry j of every span shares slot j.  The arrays are kept alive here until
        the next two batches have been staged (the library reads them until the batch has been aligned)."""
        arr = (BhipQuerySpan * max(1, len(spans)))()
        keep = []
        n_tot = 0
        for k, sp in enumerate(spans):
            codes, off, emac = _arr(sp["codes"], np.uint8), _arr(sp["off"], np.uint64), _arr(sp["emac"], np.uint16)
            rc, fl = _arr(sp.get("rc"), np.uint8), _arr(sp.get("flags"), np.uint8)
            keep += [codes, off, emac, rc, fl]
            n = len(off) - 1
            arr[k].codes, arr[k].off, arr[k].emac = codes.ctypes.data, off.ctypes.data, emac.ctypes.data
            c4 = _arr(sp.get("codes4"), np.uint8)
            keep.append(c4)
            arr[k].codes4 = c4.ctypes.data if c4 is not None else None
            arr[k].rc = rc.ctypes.data if rc is not None else None
            arr[k].flags = fl.ctypes.data if fl is not None else None
            arr[k].n, arr[k].q_base = n, int(sp.get("q_base", 0))
            c2, ln = _arr(sp.get("codes2"), np.uint8), _arr(sp.get("len"), np.uint16)
            keep += [c2, ln]
            arr[k].codes2 = c2.ctypes.data if c2 is not None else None
            arr[k].len = ln.ctypes.data if ln is not None else None
            n_tot += n
        self._span_keep = getattr(self, "_span_keep", [])[-2:] + [keep]
        self._staged_n = n_tot
        _chk(lib().bhip_stage_spans(self._h, arr, len(spans), int(n_shared), int(max_len)))

    def align_staged(self, all_hits=False, out=None):
        """out: optional preallocated HIT_DTYPE array reused between calls"""
        hits = out if out is not None else np.zeros(max(1 << 16, 4 * (self._staged.n if getattr(self, "_staged", None) is not None else getattr(self, "_staged_n", 0))), dtype=HIT_DTYPE)
        while True:
            n = C.c_uint64()
            rc = lib().bhip_align_staged(self._h, _hits_mode(all_hits), _ptr(hits), len(hits), C.byref(n))
            if rc == BHIP_E_CAPACITY:
                hits = np.zeros(int(n.value) + 16, dtype=HIT_DTYPE)
                continue
            _chk(rc)
            return hits[:n.value], hits

    def sync_hits(self):
        """with option async_d2h: wait until the records of the previous calls are in their host buffers"""
        _chk(lib().bhip_sync_hits(self._h))

    def copy_hits_device(self, dst_ptr, cap_records):
        """device-to-device copy of the last call's records into caller-owned device memory (e.g. a torch tensor's data_ptr())"""
        n = C.c_uint64()
        _chk(lib().bhip_copy_hits_device(self._h, C.c_void_p(int(dst_ptr)), int(cap_records), C.byref(n)))
        return int(n.value)

    def align_pairs(self, q, pair_q, pair_clump):
        pair_q = _arr(pair_q, np.uint32)
        pair_clump = _arr(pair_clump, np.uint32)
        mins = np.zeros(len(pair_q) * 16, dtype=np.uint8)
        _chk(lib().bhip_align_pairs(self._h, _ptr(q.codes), _ptr(q.off), _ptr(q.emac), q.n, _ptr(pair_q), _ptr(pair_clump),
                                    len(pair_q), _ptr(mins)))
        return mins.reshape(-1, 16)

    def prefilter(self, q, cap=None):
        cap = cap or max(1 << 16, 64 * q.n)
        while True:
            oq = np.zeros(cap, np.uint32)
            oc = np.zeros(cap, np.uint32)
            on = np.zeros(cap, np.uint32)
            n = C.c_uint64()
            rc = lib().bhip_prefilter(self._h, _ptr(q.codes), _ptr(q.off), _ptr(q.emac), q.n, _ptr(oq), _ptr(oc), _ptr(on), cap, C.byref(n))
            if rc == BHIP_E_CAPACITY:
                cap = int(n.value) + 16
                continue
            _chk(rc)
            k = n.value
            return oq[:k], oc[:k], on[:k]
