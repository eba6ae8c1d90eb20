#!/bin/bash
# round 4, second GPU call: kernel parity, scale-7 A/B of the register-resident record blocks, one-batch job, phase timers
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
W=/dev/shm/bb
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu > $O/r04b_tests.txt 2>&1; echo "tests exit $?" >> $O/r04b_tests.txt
tail -4 $O/r04b_tests.txt
BHIP_DEBUG=1 timeout 1500 python bench.py --workdir $W --db-scale 7 --drop-refs --no-cpu-baseline --no-end-to-end --ab prefilter_rb=2 --ab prefilter_rb=3 --ab prefilter_rb=4 --ab seed_min_need=0,prefilter_rb=4 --ab seed_min_need=0,prefilter_rb=2 > $O/r04b_bench_s7.json 2> $O/r04b_bench_s7.err
grep "^\[bench\] ab\|accelerator built\|prefilter kernel:\|^\[bench\] rank\|database built" $O/r04b_bench_s7.err | sort | uniq -c | cut -c1-400
python tools/bsum.py s7 < $O/r04b_bench_s7.json
for P in 1 4 3 6; do
  BURST_HOST_PIECES=$P timeout 600 python bench.py --workdir $W --db-scale 7 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 0 --reads 1250000 > $O/r04b_one_$P.json 2> $O/r04b_one_$P.err
  python tools/bsum.py one_batch_pieces_$P < $O/r04b_one_$P.json
done
for RB in 2 3; do
  BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 600 python bench.py --workdir $W --db-scale 7 --no-cpu-baseline --no-end-to-end --opt prefilter_rb=$RB > $O/r04b_prof_rb$RB.json 2> $O/r04b_prof_rb$RB.err
  grep "phase share" $O/r04b_prof_rb$RB.err; python tools/bsum.py prof_rb$RB < $O/r04b_prof_rb$RB.json
done
rm -rf $W
