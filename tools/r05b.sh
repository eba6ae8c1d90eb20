#!/bin/bash
# round 5: k_prefilter_cw (one query per wave) -- parity of the kernel tests with it as variant and as default, then A/B at the metric's size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
T0=$SECONDS
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/r05b_tests.txt 2>&1; echo "tests exit $? after $((SECONDS - T0)) s" >> $O/r05b_tests.txt
tail -5 $O/r05b_tests.txt
T0=$SECONDS
BHIP_OPTS=prefilter_cw=1 timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu -k "not randomised" > $O/r05b_tests_cw.txt 2>&1; echo "tests (cw default) exit $? after $((SECONDS - T0)) s" >> $O/r05b_tests_cw.txt
tail -5 $O/r05b_tests_cw.txt
S=${1:-11.37}; W=/dev/shm/bbf
C="--workdir $W --db-scale $S --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
T0=$SECONDS
BHIP_DEBUG=1 timeout 1500 python bench.py $C --ab prefilter_cw=1 --ab prefilter_cw=1,seed_min_need=0 --ab prefilter_cw=1,seed_min_need=3 --ab prefilter_cw=1,prefilter_waves=16 --ab prefilter_cw=1,prune=0 --ab prune=0 > $O/r05b_bench.json 2> $O/r05b_bench.err
echo "bench exit $? after $((SECONDS - T0)) s"
grep "^\[bench\] ab\|^\[bench\] rank\|database built\|overflowed" $O/r05b_bench.err | sort | uniq -c | sort -rn | head -24 | cut -c1-420
grep "prefilter kernel:" $O/r05b_bench.err | sort | uniq -c
python tools/bsum.py full < $O/r05b_bench.json
# the small database of rounds 1-3 (35 records per read): the other end of the range
timeout 900 python bench.py --db-scale 1 --workdir /dev/shm/bb1 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --ab prefilter_cw=1 > $O/r05b_small.json 2> $O/r05b_small.err
grep "^\[bench\] ab" $O/r05b_small.err | cut -c1-300; python tools/bsum.py small < $O/r05b_small.json
rm -rf $W /dev/shm/bb1
