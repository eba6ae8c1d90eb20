"""ctypes access to the checker: oracle/liboracle.so (our restatement) and, when present,
oracle/_ref/libref_harness.so (the real reference kernels).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORC_HIT = np.dtype([("q", "<u4"), ("refIx", "<u4"), ("finalPos", "<u4"), ("score", "<f4"),
                    ("ed", "u1"), ("gapR", "u1"), ("gapQ", "u1"), ("rc", "u1")])

_orc = None
_ref = None


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def oracle():
    global _orc
    if _orc is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "burst_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
        L = C.CDLL(so)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.orc_score_lut.argtypes = [i32, vp]
        L.orc_char2code.argtypes = [vp]
        L.orc_rc_code.argtypes = [C.c_uint8]; L.orc_rc_code.restype = C.c_uint8
        L.orc_error_budget.argtypes = [C.c_float, u32]; L.orc_error_budget.restype = u32
        L.orc_unpack_clump.argtypes = [vp, u32, vp]
        L.orc_aded_clump.argtypes = [vp, u32, vp, u32, u32, vp, vp]; L.orc_aded_clump.restype = u32
        L.orc_rescore_lane.argtypes = [vp, u32, vp, u32, u32, vp, vp]; L.orc_rescore_lane.restype = i32
        L.orc_search.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, vp, u32, u32, vp, i32, vp, u64]; L.orc_search.restype = u64
        L.orc_prefilter_counts.argtypes = [vp, u32, u32, i32, u32, vp, vp, u32, vp]; L.orc_prefilter_counts.restype = u32
        _orc = L
    return _orc


def have_reference():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_harness.so"))


def ref_binary(k=12):
    p = os.path.join(ORACLE_DIR, "_ref", "burst%d" % k)
    return p if os.path.exists(p) else None


def reference():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_harness.so"))
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.ref_setscore.argtypes = [i32]
        L.ref_get_scorefast.argtypes = [vp]
        L.ref_translate.argtypes = [vp, C.c_size_t]
        L.ref_align_clump.argtypes = [vp, u32, vp, u32, u32, i32, u32, vp, i32, i32, vp, vp, vp, vp]
        L.ref_align_clump.restype = u32
        _ref = L
    return _ref


def score_lut(z=1):
    lut = np.zeros(256, np.uint8)
    oracle().orc_score_lut(z, _p(lut))
    return lut


def unpack_clump(packed, clump_len):
    rows = np.zeros((clump_len, 16), np.uint8)
    oracle().orc_unpack_clump(_p(packed), clump_len, _p(rows))
    return rows


def aded_clump(rows, q, max_ed, lut):
    rows = np.ascontiguousarray(rows, np.uint8); q = np.ascontiguousarray(q, np.uint8)
    mins = np.zeros(16, np.uint8)
    ret = oracle().orc_aded_clump(_p(rows), rows.shape[0], _p(q), len(q), max_ed, _p(lut), _p(mins))
    return ret, mins


def rescore_lane(q, r, bound, lut):
    q = np.ascontiguousarray(q, np.uint8); r = np.ascontiguousarray(r, np.uint8)
    out = np.zeros(1, ORC_HIT)
    ok = oracle().orc_rescore_lane(_p(q), len(q), _p(r), len(r), bound, _p(lut), _p(out))
    return bool(ok), out[0]


def search(packed, clump_len, tot_refs, qcodes, qoff, qE, qsix, qrc, n_shared, lut, all_hits, cap=1 << 20):
    clump_len = np.ascontiguousarray(clump_len, np.uint32)
    qE = np.ascontiguousarray(qE, np.uint32); qsix = np.ascontiguousarray(qsix, np.uint32)
    qrc = np.ascontiguousarray(qrc, np.uint8); qoff = np.ascontiguousarray(qoff, np.uint64)
    hits = np.zeros(cap, ORC_HIT)
    n = oracle().orc_search(_p(packed), _p(clump_len), len(clump_len), tot_refs, _p(qcodes), _p(qoff), _p(qE), _p(qsix), _p(qrc),
                            len(qE), n_shared, _p(lut), int(all_hits), _p(hits), cap)
    assert n <= cap
    return hits[:n]


def prefilter_counts(q, E, K, offs, entries, n_clumps, stride=1):
    q = np.ascontiguousarray(q, np.uint8)
    counts = np.zeros(n_clumps, np.uint16)
    n = oracle().orc_prefilter_counts(_p(q), len(q), E, K, stride, _p(offs), _p(entries), n_clumps, _p(counts))
    return n, counts


def ref_align_clump(rows, q, max_ed, variant=0, minlen=None, rescore=True, bound_override=-1):
    """the real reference on one (query, clump): returns ret, mins[16], (score, finalPos, gapR, gapQ) arrays"""
    R = reference()
    rows = np.ascontiguousarray(rows, np.uint8); q = np.ascontiguousarray(q, np.uint8)
    mins = np.zeros(16, np.uint8); score = np.zeros(16, np.float32); fin = np.zeros(16, np.uint32)
    gr = np.zeros(16, np.uint8); gq = np.zeros(16, np.uint8)
    ret = R.ref_align_clump(_p(rows), rows.shape[0], _p(q), len(q), max_ed, variant, len(q) if minlen is None else minlen,
                            _p(mins), int(rescore), bound_override, _p(score), _p(fin), _p(gr), _p(gq))
    return ret, mins, score, fin, gr, gq
