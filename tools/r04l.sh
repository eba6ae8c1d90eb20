#!/bin/bash
# round 4: kernel parity with byte counters, then A/B at the default database size
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu > $O/r04l_tests.txt 2>&1; echo "tests exit $?" >> $O/r04l_tests.txt
tail -4 $O/r04l_tests.txt
BHIP_DEBUG=1 timeout 1500 python bench.py --db-scale 7 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job --ab prefilter_bytes=0 --ab prefilter_bytes=0,seed_min_need=0 --ab seed_min_need=0 --ab seed_min_need=2 > $O/r04l_bench.json 2> $O/r04l_bench.err
echo "bench exit $?"
grep "^\[bench\] ab\|overflowed" $O/r04l_bench.err | sort | uniq -c | sort -rn | head -12 | cut -c1-300
python tools/bsum.py s7 < $O/r04l_bench.json
