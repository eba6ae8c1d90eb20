#!/bin/bash
# round 4: kernel parity after the barrier / second-pass changes, then A/B at the metric's database size (one database, three processes)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
S=${1:-11.37}; W=/dev/shm/bbf
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu > $O/r04f_tests.txt 2>&1; echo "tests exit $?" >> $O/r04f_tests.txt
tail -4 $O/r04f_tests.txt
C="--workdir $W --db-scale $S --keep-files --no-cpu-baseline --no-end-to-end --no-continuity"
BHIP_DEBUG=1 timeout 1500 python bench.py $C --ab prefilter_table=10 --ab seed_min_need=0 --ab seed_min_need=0,prefilter_table=10 --ab prefilter_rb=3 --ab prefilter_rb=2 > $O/r04f_bench.json 2> $O/r04f_bench.err
echo "bench exit $?"
grep "^\[bench\] ab\|accelerator built\|^\[bench\] rank\|database built\|overflowed" $O/r04f_bench.err | sort | uniq -c | sort -rn | head -24 | cut -c1-420
grep "prefilter kernel:" $O/r04f_bench.err | sort | uniq -c
python tools/bsum.py full < $O/r04f_bench.json
BURST_AMD_LIBDIR=$R/burst_amd/barriers timeout 900 python bench.py $C > $O/r04f_barriers.json 2> $O/r04f_barriers.err
python tools/bsum.py with_barriers < $O/r04f_barriers.json
BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 900 python bench.py $C > $O/r04f_prof.json 2> $O/r04f_prof.err
grep "phase share" $O/r04f_prof.err; python tools/bsum.py prof < $O/r04f_prof.json
rm -rf $W
