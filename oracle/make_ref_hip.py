#!/usr/bin/env python3
"""oracle/make_ref_hip.py -- TEST INFRASTRUCTURE, build container only.  Compile-and-link check of the reference-side binding
(INTEGRATION.md): splices oracle/burst_hip_binding.inc into a SCRATCH copy of /root/reference/burst.c (in /tmp; nothing of
the reference enters the repository), compiles it with -DBURST_HIP against include/burst_hip.h and links it with
burst_amd/libburst_hip.so.  Output: oracle/_ref/burst12_hip and burst15_hip (git-ignored; they travel to the GPU box, where
tests/test_gpu_e2e.py runs the reference's own host around our device path against the golden .b6 files).

  python oracle/make_ref_hip.py [/root/reference]
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "burst.c")
    if not os.path.exists(src):
        print("make_ref_hip: %s absent, keeping prebuilt oracle/_ref" % src)
        return 0
    lib = os.path.join(ROOT, "burst_amd", "libburst_hip.so")
    if not os.path.exists(lib):
        print("make_ref_hip: %s not built yet" % lib)
        return 1
    lines = open(src, encoding="latin-1").read().split("\n")
    # anchor 1: the function the binding lives in; the header goes in front of it
    fn = [i for i, ln in enumerate(lines) if re.match(r"\s*static inline void do_alignments\(", ln)]
    # anchor 2: the accelerated branch, first `if (DO_ACCEL) {` at column 0 after the BasePod declaration
    bp = [i for i, ln in enumerate(lines) if "**BasePod = 0;" in ln]
    assert len(fn) == 1 and len(bp) == 1, "reference layout changed"
    acc = [i for i in range(bp[0], len(lines)) if lines[i].startswith("if (DO_ACCEL) {")]
    eoa = [i for i, ln in enumerate(lines) if re.match(r"\s*EOA:", ln)]
    assert acc and len(eoa) == 1 and eoa[0] > acc[0], "reference layout changed"
    binding = open(os.path.join(HERE, "burst_hip_binding.inc")).read().split("\n")
    out = lines[:fn[0]] + ["#ifdef BURST_HIP", '#include "burst_hip.h"', "#endif"] + lines[fn[0]:acc[0]] + binding + lines[acc[0]:]
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="burst_hip_binding_") as tmp:
        patched = os.path.join(tmp, "burst_patched.c")
        open(patched, "w", encoding="latin-1").write("\n".join(out))
        for K in (12, 15):
            exe = os.path.join(HERE, "_ref", "burst%d_hip" % K)
            cmd = ["gcc", "-std=gnu11", "-O3", "-march=x86-64-v3", "-fopenmp", "-w", "-DSCOUR_N=%d" % K, "-DBURST_HIP", "-I" + os.path.join(ROOT, "include"),
                   patched, "-o", exe, "-L" + os.path.join(ROOT, "burst_amd"), "-lburst_hip", "-Wl,-rpath,$ORIGIN/../../burst_amd", "-Wl,--allow-shlib-undefined", "-lm"]
            subprocess.check_call(cmd)
            syms = subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True)
            for s in ("bhip_init", "bhip_align_batch", "bhip_destroy", "bhip_last_error"):
                assert s in syms, "%s does not reference %s" % (exe, s)
            print("make_ref_hip: built %s (binding compiled and linked against libburst_hip.so)" % exe)
    return 0


if __name__ == "__main__":
    sys.exit(main())
