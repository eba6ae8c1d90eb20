"""The C-ABI library loads and exports every symbol include/burst_hip.h declares (no compute without a GPU),
and refuses to work without a device instead of falling back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "burst_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bhip_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from burst_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert set(capi.EXPORTS) == set(syms)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.bhip_abi_version() == 8
    assert ctypes.sizeof(capi.BhipStats) == 12 * 8 + 13 * 4 + 4 * 4 + 4      # 12 u64 + 13 f32 + 4 u32, padded to 8
    assert capi.HIT_DTYPE.itemsize == 20


def test_host_library_exports():
    lib = ctypes.CDLL(os.path.join(ROOT, "burst_amd", "libburst_host.so"))
    for s in ["bh_queries_load", "bh_edx_read", "bh_acx_read", "bh_db_from_fasta", "bh_align", "bh_report", "bh_report_ex", "bh_device_open",
              "bh_acx_build", "bh_edx_write", "bh_score_lut", "bh_error_budget"]:
        assert hasattr(lib, s), s


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_without_device():
    from burst_amd import capi
    lut = np.zeros(256, np.uint8)
    with pytest.raises(capi.BurstHipError) as e:
        capi.Device(np.zeros(64, np.uint8), np.array([8], np.uint32), 1, lut)
    assert e.value.code == capi.BHIP_E_DEVICE


def test_argument_validation():
    from burst_amd import capi
    with pytest.raises(capi.BurstHipError) as e:
        capi.Device(np.zeros(64, np.uint8), np.array([8], np.uint32), 1, np.zeros(256, np.uint8), xalpha=1)
    assert e.value.code == capi.BHIP_E_ARG


def test_lane_set_codes_cover_every_mask(tmp_path):
    """the top byte of a 4-byte accelerator record (burst_amd/csrc/bhip_lanecode.h): every 16-bit lane mask gets the code of a
    superset, exact for one and two lanes, the smallest superset any code offers beyond; 167 codes, no two alike"""
    import ctypes as C
    import subprocess
    import numpy as np
    src = tmp_path / "lc.c"
    src.write_text('#include "bhip_lanecode.h"\n'
                   'unsigned enc(unsigned m) { return bhip_lane_mask_code(m); }\n'
                   'unsigned dec(unsigned c) { return bhip_lane_code_mask(c); }\n')
    so = tmp_path / "lc.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "burst_amd", "csrc"), str(src), "-o", str(so)])
    L = C.CDLL(str(so))
    L.enc.restype = L.dec.restype = C.c_uint
    dec = np.array([L.dec(c) for c in range(256)], np.uint32)
    assert len(set(dec[:167].tolist())) == 167 and dec[166] == 0xFFFF and np.all(dec[167:] == 0xFFFF)
    pc = np.array([bin(x).count("1") for x in range(1 << 16)], np.uint32)
    masks = np.arange(1, 1 << 16, dtype=np.uint32)
    codes = np.array([L.enc(int(m)) for m in masks], np.uint32)
    assert codes.max() == 166 and L.enc(0) == 166
    got = dec[codes]
    assert np.all(got & masks == masks)
    assert np.all(got[pc[masks] <= 2] == masks[pc[masks] <= 2])
    # no code names a smaller superset
    best = np.full(len(masks), 17, np.uint32)
    for c in range(167):
        ok = (dec[c] & masks) == masks
        best[ok] = np.minimum(best[ok], pc[dec[c]])
    assert np.array_equal(pc[got], best)
