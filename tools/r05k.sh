#!/bin/bash
# round 5: k_hit_fix as a wave-per-group rank sort -- the kernel tests with groups of 29..333 records, the fuzzer, the line at the metric's
# size (must not move) and the strains line (21.8 records per read) with the reference beside it
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1200 TAG=r05k bash tools/run.sh tests tests/test_gpu_kernels.py
timeout 600 python tests/fuzz_gpu.py 120 47 > $O/r05k_fuzz.txt 2>&1; tail -2 $O/r05k_fuzz.txt
W=/dev/shm/burst_amd_bench
BHIP_DEBUG=1 python bench.py --workdir $W --db-scale 11.37 --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job > $O/r05k_bench.json 2> $O/r05k_bench.err
grep "accelerator built\|^\[bench\] rank" $O/r05k_bench.err | cut -c1-400; python tools/bsum.py r05k < $O/r05k_bench.json
rm -rf $W
TAG=r05k_strains bash tools/run.sh bench --workdir /dev/shm/bb_strains --db-profile strains --no-continuity --no-end-to-end
rm -rf /dev/shm/bb_strains
