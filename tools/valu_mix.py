"""Static VALU instruction mix of the hot kernels: python tools/valu_mix.py [out.json]
Disassembles burst_amd/csrc/bhip_kernels.hip for gfx950 and, per kernel, counts the VALU instructions by issue class.  Issue
costs are the ones tools/ubench/valu_rate.hip measures on MI355X (cycles per wave64 instruction per SIMD at the clock the chip
holds under load, ~2.05 GHz): 2 for the VOP1/VOP2 forms and v_bitop3_b32, 4 for every other VOP3 form, v_addc/v_subb with carry,
v_lshlrev_b32 and DPP moves.  valu_frac in profiles/pmc_summary.json counts every instruction as 2 cycles at 2.4 GHz; multiplied by
`weight` (mean cycles / 2 x 2.4 / 2.05) it becomes the share of the VALU issue capacity the kernel really uses.  The mix is static
(whole kernel body); the hot kernels are dominated by their unrolled inner loops."""
import json, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"k_myers_prefix_task<1>": "_Z19k_myers_prefix_taskILi1E", "k_myers_window<4>": "_Z14k_myers_windowILi4E", "k_myers_window_band<2>": "_Z19k_myers_window_bandILi2E", "k_rescore_reg<0>": "_Z13k_rescore_regILi0E",
           "k_prefilter_cf<9>": "_Z14k_prefilter_cfILi9E", "k_seed_ranges": "_Z13k_seed_ranges", "k_build_peq": "_Z11k_build_peq"}
HALF_E32 = ("v_lshlrev_b32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_mul_")
def cost(m):
    if m.startswith("v_bitop3"): return 2
    if "_dpp" in m or "_sdwa" in m: return 4
    if m.endswith("_e64") or any(m.startswith(h) for h in HALF_E32): return 4
    if m.endswith("_e32"): return 2
    return 4          # VOP3-only opcodes carry no suffix (v_min3, v_and_or, v_alignbit, v_lshl_or, v_add3, v_mad, v_bfe, v_perm ...)
def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "valu_mix.json")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(R, "include"), "-S", "--cuda-device-only",
                               os.path.join(R, "burst_amd/csrc/bhip_kernels.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    res = {}
    for name, sym in KERNELS.items():
        start = next((i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or (l.startswith(sym) and ": " in l and l.split(":")[0].startswith(sym))), None)
        if start is None: continue
        n = {2: 0, 4: 0}; nops = 0; lds = 0; salu = 0; vmem = 0; top = {}
        for l in lines[start + 1:]:
            t = l.strip()
            if t.startswith("s_endpgm"): break
            m = re.match(r"([a-z_0-9]+)", t)
            if not m: continue
            op = m.group(1)
            if op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane") and not op.startswith("v_writelane"):
                c = cost(op); n[c] += 1; top[op] = top.get(op, 0) + 1
            elif op == "s_nop": nops += 1 + int(re.search(r"s_nop (\d+)", t).group(1))
            elif op.startswith("ds_"): lds += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): vmem += 1
            elif op.startswith("s_"): salu += 1
        tot = n[2] + n[4]
        if not tot: continue
        mean = (2.0 * n[2] + 4.0 * n[4]) / tot
        res[name] = {"valu_static": tot, "full_rate": n[2], "half_rate": n[4], "mean_issue_cycles": mean, "weight": mean / 2.0 * 2.4 / 2.05,
                     "s_nop_wait_states": nops, "lds": lds, "vmem": vmem, "salu": salu,
                     "most_frequent": sorted(top.items(), key=lambda kv: -kv[1])[:8]}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-26s VALU %6d (full %6d, half %6d) mean %.2f cycles, weight %.2f, s_nop states %d, LDS %d, VMEM %d" % (k, v["valu_static"], v["full_rate"], v["half_rate"], v["mean_issue_cycles"], v["weight"], v["s_nop_wait_states"], v["lds"], v["vmem"]))
if __name__ == "__main__":
    main()
