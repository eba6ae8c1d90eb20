#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box: tools/profile_round.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_{kernel_stats.txt,pmc.txt,traffic.json,bench.json}; copy what should be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; TAG=$1; shift
O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --no-strains --keep-files --no-prime $*"      # --no-prime: every dispatch of a batch kernel is a full-size batch; --keep-files: the five passes share the database
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- $B > $O/${TAG}_kt.log 2>&1
grep '^{' $O/${TAG}_kt.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -- $B > $O/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${TAG}_sq -- $B > $O/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/${TAG}_sq2 -- $B > $O/${TAG}_sq2.log 2>&1
python - "$O" "$TAG" "$B" <<'PY'
import csv, glob, sys, json, collections
O, TAG, CMD = sys.argv[1:4]
def short(n): return n.replace('HIP_vector_type<unsigned int, 2u>', 'uint2').split('(')[0][:48]
out = []
for f in glob.glob('%s/%s_kt/**/*kernel_stats.csv' % (O, TAG), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    out.append('command: rocprofv3 --kernel-trace --stats -- %s' % CMD)
    out.append('%-50s %7s %12s %12s %7s' % ('kernel', 'calls', 'avg_us', 'total_ms', 'share'))
    for r in rows[:70]:
        out.append('%-50s %7s %12.1f %12.3f %6.1f%%' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
open('%s/%s_kernel_stats.txt' % (O, TAG), 'w').write('\n'.join(out) + '\n')
def pmc(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for f in glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, TAG, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name']); agg[k][r['Counter_Name']] += float(r['Counter_Value']); calls[k].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in calls.items()}
fa, fc = pmc('fetch'); wa, wc = pmc('write'); sa, sc = pmc('sq'); s2, _ = pmc('sq2')
lines = ['command: rocprofv3 --pmc <set> --kernel-trace -- %s   (one pass per counter set; totals over all dispatches of the run)' % CMD]
traffic = {}
for k in sorted(fa, key=lambda k: -fa[k].get('FETCH_SIZE', 0)):
    fetch_kb = fa[k].get('FETCH_SIZE', 0.0); write_kb = wa.get(k, {}).get('WRITE_SIZE', 0.0); n = max(1, fc.get(k, 1))
    hbm = (2.0 * fetch_kb * 1024 + write_kb * 1024) / n
    lines.append('%-50s dispatches %5d FETCH_SIZE_KB %14.1f WRITE_SIZE_KB %14.1f  hbm_bytes/launch (2*fetch+write) %14.0f' % (k, n, fetch_kb, write_kb, hbm))
    if k.replace('void ', '').split('<')[0] in traffic: continue      # template variants share a name: sorted by bytes, the heaviest is first
    traffic[k.replace('void ', '').split('<')[0]] = {"kernel": k, "dispatches": n, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb, "hbm_bytes_per_launch": hbm,
        "hbm_bytes_per_launch_gather_calibrated": (fetch_kb * 1024 + write_kb * 1024) / n,
        "gather_note": "FETCH_SIZE x 1024 = 64 B x the distinct 128-byte lines a dispatch requests, whatever the pattern (profiles/r06e_fetch_calibration_runs.txt, tools/calibrate_fetch.sh): hbm_bytes_per_launch = every requested line moved whole (upper bound), this figure = one 64-byte half per requested line (lower bound)",
        "correction": "gfx950: FETCH_SIZE tallies a 128-byte line request as 64 B (MI355X_MICROARCH.md, HBM section; calibrated on runs of 1 / 16 / 40 / 164 dwords in round 6); WRITE_SIZE as reported"}
lines.append('')
for k in sorted(sa, key=lambda k: -sa[k].get('SQ_WAVE_CYCLES', 0)):
    v = dict(sa[k]); v.update(s2.get(k, {})); wc_ = max(1.0, v.get('SQ_WAVE_CYCLES', 1.0))
    lines.append('%-50s ' % k + ' '.join('%s=%.4g' % (c, x) for c, x in sorted(v.items())) + '  | active %.0f%% wait_any %.0f%% wait_inst %.0f%%' % (100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc_, 100 * v.get('SQ_WAIT_ANY', 0) / wc_, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc_))
open('%s/%s_pmc.txt' % (O, TAG), 'w').write('\n'.join(lines) + '\n')
json.dump(traffic, open('%s/%s_traffic.json' % (O, TAG), 'w'), indent=1)
# per-kernel summary bench.py reads (profiles/pmc_summary.json): HBM bytes per launch and the share of the VALU issue peak
# (wave-level VALU instructions x 2 cycles on a SIMD32, against 1024 SIMDs x 2.4 GHz over the kernel's average duration)
avg_ns = {}
for f in glob.glob('%s/%s_kt/**/*kernel_stats.csv' % (O, TAG), recursive=True):
    for r in csv.DictReader(open(f)):
        avg_ns[short(r['Name'])] = float(r['AverageNs'])
summary = {}
def weight(k):
    return avg_ns.get(k, 0.0) * max(1, sc.get(k, fc.get(k, 1)))
for k in sorted(set(list(fa) + list(sa)), key=weight):      # template variants share a name: the one with the most time wins
    name = k.replace('void ', '').split('<')[0]
    e = summary[name] = {"kernel": k}
    if k in fa and traffic.get(name, {}).get("kernel") == k:
        e.update(traffic[name])
    if k in sa:
        n = max(1, sc.get(k, 1)); insts = sa[k].get('SQ_INSTS_VALU', 0.0) / n; dur = avg_ns.get(k)
        e["valu_insts_per_launch"] = insts
        if dur:
            e["avg_ns"] = dur; e["valu_frac"] = insts * 2.0 / (dur * 1e-9 * 1024 * 2.4e9)
        wc_ = max(1.0, sa[k].get('SQ_WAVE_CYCLES', 1.0))
        e["issue_stall_frac"] = sa[k].get('SQ_WAIT_INST_ANY', 0.0) / wc_; e["wait_frac"] = sa[k].get('SQ_WAIT_ANY', 0.0) / wc_
        v2 = s2.get(k, {})
        if v2.get('SQ_LDS_IDX_ACTIVE'):
            e["lds_bank_conflict_frac"] = v2.get('SQ_LDS_BANK_CONFLICT', 0.0) / v2['SQ_LDS_IDX_ACTIVE']
import os
summary["_meta"] = {"tag": TAG, "commit": os.environ.get("PROFILE_COMMIT", "unknown"), "command": CMD, "record_bytes": 4,
                    "kernels": sorted(set(avg_ns)), "made_by": "tools/profile_round.sh (five rocprofv3 passes: --kernel-trace --stats, then one --pmc pass per counter set)"}
json.dump(summary, open('%s/%s_pmc_summary.json' % (O, TAG), 'w'), indent=1)
print('\n'.join(out[:16])); print('\n'.join(lines[:8]))
PY
# the raw rocprofv3 output is tens of megabytes per pass (gpurun brings back at most 64 MiB): keep the kernel trace of the first pass (the
# batch timeline is made from it, tools/timeline.py) compressed, drop the rest
python $R/tools/timeline.py $O/${TAG}_kt 2 > $O/${TAG}_batch_timeline.txt 2>&1
for f in $(find $O/${TAG}_kt -name '*kernel_trace.csv' | head -1); do grep -v "rocprim\|k_acx\|fillBuffer" $f | gzip -c > $O/${TAG}_kernel_trace.csv.gz; done
rm -rf $O/${TAG}_kt $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq $O/${TAG}_sq2
