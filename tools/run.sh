#!/bin/bash
# One parametrised recipe file for the GPU box (replaces the per-call scripts of rounds 1-4):  gpurun -- 'bash tools/run.sh <what> [args]'
# Everything lands in gpurun_out/<TAG>_* (TAG from the environment, default "run"); copy what should be judged into profiles/.
#   tests [pytest args]          the -m gpu suite (default: all of tests/), every failure listed
#   bench [bench.py args]        the driver's command line (python bench.py --gpus 1 --steps 20 --warmup 5) with the job's memory sampled beside it
#   ab <db-scale> <spec> ...     bench.py --ab <spec> ... on one resident database (spec = name=value[,name=value]; library options of bhip_set_option)
#   variant <libdir> [args]      bench.py with the libraries of burst_amd/<libdir> (tools/build_variant.sh: prof = phase timers, ...)
#   benchprof [bench.py args]    the driver's command line under rocprofv3 --kernel-trace --stats: bench line + kernel statistics of one process
#   profile [bench.py args]      tools/profile_round.sh: rocprofv3 --kernel-trace --stats + the PMC passes -> kernel_stats / pmc / pmc_summary
#   shape configs1|configs2|configs4   the other BASELINE shapes on their own databases, one bench line each with the reference beside it
#   cli <db-scale> [burst_hip args]    the burst_hip command line on the bench's database and read pool, BHIP_DEBUG phase lines kept
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${TAG:-run}; WHAT=$1; shift
W=${WORKDIR:-/dev/shm/burst_amd_bench}
QUIET="--no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --no-strains"
sum() { python tools/bsum.py "$1" < "$2"; }
case "$WHAT" in
tests)
	T0=$SECONDS
	timeout ${LIMIT:-2400} python -m pytest "${@:-tests}" -q -m gpu --durations=40 > $O/${TAG}_gputests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/${TAG}_gputests.txt
	grep -n "^FAILED\|^ERROR\|passed\|failed\|tests exit" $O/${TAG}_gputests.txt | tail -12 ;;
bench)
	(while true; do echo "$(date +%s) $(cat /sys/fs/cgroup/memory.current 2>/dev/null) $(df --output=used -B1 /dev/shm | tail -1) $(grep -E '^(anon|file|shmem|file_mapped|kernel) ' /sys/fs/cgroup/memory.stat 2>/dev/null | tr '\n' ' ')"; sleep 2; done) > $O/${TAG}_mem.txt & MW=$!
	T0=$SECONDS
	timeout ${LIMIT:-2400} python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
	echo "bench exit $? after $((SECONDS - T0)) s"; kill $MW
	grep "^\[bench\]" $O/${TAG}_bench.err | cut -c1-330; sum "$TAG" $O/${TAG}_bench.json
	python - "$O/${TAG}_bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for k in ("cpu_baseline", "cpu_baseline_skipped", "parity_vs_reference", "gpu_over_cpu", "end_to_end"): print(k, json.dumps(d.get(k))[:700])
print("at metric size:", d["config"]["extrapolation"]["this_run_is_at_metric_size"], "| roofline", d["roofline"]["kernel"], "%.4f" % d["roofline"]["frac"])
PY
	echo "peak memory: $(sort -k2 -n $O/${TAG}_mem.txt | tail -1)" ;;
benchprof)
	# the driver's command line UNDER rocprofv3 --kernel-trace --stats: the bench line (HIP events) and the kernel statistics of ONE process,
	# so that roofline.frac can be recomputed from the profiler's average of the same launches (round 5's verdict: 9 % between two runs)
	cd /tmp && export TMPDIR=/tmp
	T0=$SECONDS
	timeout ${LIMIT:-2400} rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --db-scale 11.37 --no-end-to-end --no-continuity --no-strains --no-short-job "$@" > $O/${TAG}_benchprof.log 2> $O/${TAG}_benchprof.err
	echo "benchprof exit $? after $((SECONDS - T0)) s"
	grep '^{' $O/${TAG}_benchprof.log | tail -1 > $O/${TAG}_bench_under_rocprof_driver_line.json
	python - "$O" "$TAG" <<'PY'
import csv, glob, json, sys, os
O, TAG = sys.argv[1:3]
best = None
for f in glob.glob('%s/%s_kt/**/*kernel_stats.csv' % (O, TAG), recursive=True):      # (child processes -- the reference is no HIP program -- leave none; the bench's own is the one with the prefilter kernel)
    rows = list(csv.DictReader(open(f)))
    if any('k_prefilter_cq' in r['Name'] for r in rows) and (best is None or len(rows) > len(best)): best = rows
d = json.loads(open('%s/%s_bench_under_rocprof_driver_line.json' % (O, TAG)).read())
out = ['command: rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --db-scale 11.37 --no-end-to-end --no-continuity --no-strains --no-short-job   (same process as %s_bench_under_rocprof_driver_line.json)' % TAG,
       '%-50s %7s %12s %12s' % ('kernel', 'calls', 'avg_us', 'total_ms')]
for r in (best or [])[:60]:
    n = r['Name'].replace('HIP_vector_type<unsigned int, 2u>', 'uint2').split('(')[0][:48]
    if n.startswith(('void rocprim', 'k_acx', '__amd_rocclr')): continue
    out.append('%-50s %7s %12.1f %12.3f' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
ro = d['roofline']
# the profiler's average of the dominant kernel over the FULL-SIZE launches of this process (the priming call and the 600 000-read reference sample
# launch the same kernel on smaller batches: dispatches shorter than 3/4 of the longest are left out), from the per-dispatch trace
avg, n_full, n_all = None, 0, 0
for f in glob.glob('%s/%s_kt/**/*kernel_trace.csv' % (O, TAG), recursive=True):
    du = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if ro['kernel'].split('<')[0] in r['Kernel_Name'] and ('<0, 0>' in r['Kernel_Name'] or '<' not in ro['kernel'])]
    if du and len(du) > n_all:
        ref = sorted(du, reverse=True)[min(len(du) - 1, 7)]      # (the 8th longest: a few dispatches are stretched by whatever ran beside them)
        full = sorted(x for x in du if 0.6 * ref <= x)
        med = full[len(full) // 2]
        full = [x for x in full if x <= 1.25 * med]                # ... and are left out of the average
        avg, n_full, n_all = sum(full) / len(full), len(full), len(du)
out.append('')
out.append('bench line of this process: value %.1f M reads/s, dominant kernel %s: %.1f us per launch by HIP events in the pipeline, frac %.4f' % (d['value'] / 1e6, ro['kernel'], ro['ms_per_launch'] * 1e3, ro['frac']))
if avg:
    out.append('rocprofv3 average of the same kernel over the %d full-size launches of the same process (of %d dispatches): %.1f us -> frac %.4f' % (n_full, n_all, avg / 1e3, ro['algorithmic_bytes_per_launch'] / (avg * 1e-9) / 1e9 / 8000.0))
open('%s/%s_kernel_stats_driver_line.txt' % (O, TAG), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-3:]))
PY
	rm -rf $O/${TAG}_kt ;;
ab)
	S=$1; shift; AB=""; for x in "$@"; do AB="$AB --ab $x"; done
	BHIP_DEBUG=1 timeout ${LIMIT:-1500} python bench.py --workdir $W --db-scale $S --keep-files $QUIET $AB > $O/${TAG}_ab.json 2> $O/${TAG}_ab.err
	echo "ab exit $?"; grep "^\[bench\] ab\|^\[bench\] rank\|accelerator built" $O/${TAG}_ab.err | cut -c1-420; grep "prefilter kernel:" $O/${TAG}_ab.err | sort | uniq -c; sum "$TAG" $O/${TAG}_ab.json ;;
variant)
	V=$1; shift
	BURST_AMD_LIBDIR=$R/burst_amd/$V BHIP_PROF=1 timeout ${LIMIT:-900} python bench.py --workdir $W --keep-files $QUIET "$@" > $O/${TAG}_$V.json 2> $O/${TAG}_$V.err
	grep "phase share" $O/${TAG}_$V.err; sum "$TAG/$V" $O/${TAG}_$V.json ;;
profile)
	PROFILE_COMMIT=${COMMIT:-unknown} bash tools/profile_round.sh $TAG --workdir $W "$@" 2>&1 | grep -v "rocprim\|k_acx\|fillBuffer\|k_qs_" | head -24 ;;
shape)
	case "$1" in
	configs1) A="--K 12 --n-base 3300 --n-variants 30 --db-scale 1 --mode CAPITALIST --id 0.97 --cpu-sample 300000" ;;
	configs2) A="--K 12 --n-base 3300 --n-variants 30 --db-scale 1 --mode ALLPATHS --id 0.97 --read-len 292 --edits 0,2,4,8 --cpu-sample 60000" ;;
	configs4) A="--db-scale 5 --read-len 320 --mode FORAGE --id 0.95 --fr --iupac 0.001 --edits 0,2,4,8,12 --reads 500000 --cpu-sample ${CPU_SAMPLE:-20000}" ;;
	*) echo "shape configs1|configs2|configs4"; exit 1 ;;
	esac
	timeout ${LIMIT:-2400} python bench.py --workdir $W.$1 --no-end-to-end --no-continuity --no-short-job --no-strains $A ${SHAPE_EXTRA:-} > $O/${TAG}_$1.json 2> $O/${TAG}_$1.err
	echo "$1 exit $?"; sum "$1" $O/${TAG}_$1.json; rm -rf $W.$1 ;;
cli)
	S=$1; shift
	if ! ls $W/db_*.edx > /dev/null 2>&1; then python bench.py --workdir $W --db-scale $S --keep-files $QUIET --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_cli_setup.err; fi
	EDX=$(ls $W/db_*.edx | head -1); RD=$(ls $W/reads_*.fa | head -1)
	BHIP_DEBUG=1 BURST_HOST_DEBUG=1 timeout ${LIMIT:-900} burst_amd/burst_hip -r $EDX -ad -k 15 -q $RD -o $W/cli.b6 -m BEST -i 0.98 "$@" > $O/${TAG}_cli.txt 2>&1
	echo "cli exit $?"; grep "Alignment time\| s\]\|accelerator built\|record area" $O/${TAG}_cli.txt | cut -c1-300; rm -f $W/cli.b6 ;;
*) sed -n 2,12p "$0"; exit 1 ;;
esac
