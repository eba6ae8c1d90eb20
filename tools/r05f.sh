#!/bin/bash
# round 5: k_prefilter_cq v2 (loads of the whole quad first, slot-parallel emit): parity, phase shares, A/B at both sizes
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
BHIP_OPTS=prefilter_cw=2 timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "not randomised" > $O/r05f_tests_cq.txt 2>&1; echo "tests (cq default) exit $? after $((SECONDS - T0)) s" >> $O/r05f_tests_cq.txt
tail -3 $O/r05f_tests_cq.txt
timeout 600 python tests/fuzz_gpu.py 100 31 > $O/r05f_fuzz.txt 2>&1; tail -2 $O/r05f_fuzz.txt
C1="--db-scale 1 --workdir /dev/shm/bb1 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
timeout 600 python bench.py $C1 --ab prefilter_cw=2 > $O/r05f_small.json 2> $O/r05f_small.err
grep "^\[bench\] ab" $O/r05f_small.err | cut -c1-250
BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 600 python bench.py $C1 --opt prefilter_cw=2 > $O/r05f_prof_small.json 2> $O/r05f_prof_small.err
grep "phase share" $O/r05f_prof_small.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/r05f_pmc2 -- python $R/bench.py $C1 --steps 5 --warmup 1 --no-prime --opt prefilter_cw=2 > $O/r05f_pmc2.log 2>&1
python - "$O/r05f_pmc2" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        if 'prefilter_cq<0, 0' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); calls[k].add(r['Dispatch_Id'])
for k, v in agg.items():
    n = len(calls[k])
    print('%-42s launches %d  per launch: ' % (k, n) + ' '.join('%s=%.3g' % (c, x / n) for c, x in sorted(v.items())))
PY
cd $R
rm -rf /dev/shm/bb1 $O/r05f_pmc2
S=11.37; W=/dev/shm/bbf
C="--workdir $W --db-scale $S --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
BHIP_DEBUG=1 timeout 1500 python bench.py $C --ab prefilter_cw=2 --ab prefilter_cw=2,seed_min_need=0 --ab prefilter_cw=2,prefilter_waves=12 > $O/r05f_bench.json 2> $O/r05f_bench.err
echo "bench exit $?"
grep "^\[bench\] ab\|^\[bench\] rank" $O/r05f_bench.err | cut -c1-420
grep "prefilter kernel:" $O/r05f_bench.err | sort | uniq -c
BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 900 python bench.py $C --opt prefilter_cw=2 > $O/r05f_prof_full.json 2> $O/r05f_prof_full.err
grep "phase share" $O/r05f_prof_full.err
rm -rf $W
