#!/bin/bash
# One parametrised recipe file for the GPU box (replaces the per-call scripts of rounds 1-4):  gpurun -- 'bash tools/run.sh <what> [args]'
# Everything lands in gpurun_out/<TAG>_* (TAG from the environment, default "run"); copy what should be judged into profiles/.
#   tests [pytest args]          the -m gpu suite (default: all of tests/), every failure listed
#   bench [bench.py args]        the driver's command line (python bench.py --gpus 1 --steps 20 --warmup 5) with the job's memory sampled beside it
#   ab <db-scale> <spec> ...     bench.py --ab <spec> ... on one resident database (spec = name=value[,name=value]; library options of bhip_set_option)
#   variant <libdir> [args]      bench.py with the libraries of burst_amd/<libdir> (tools/build_variant.sh: prof = phase timers, ...)
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
