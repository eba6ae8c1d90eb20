#!/bin/bash
# round 5: the accelerator build with coalesced tuple extraction and wave-aggregated list lengths (tests, then its time at the metric's size),
# the re-scorers' cost of a pad column taken from the table (-x), whether a command line that follows another process pays for that
# process's memory, the strains line with its record buffer sized ahead
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1500 TAG=r05j bash tools/run.sh tests tests/test_gpu_kernels.py tests/test_gpu_acx.py tests/test_gpu_e2e.py
W=/dev/shm/burst_amd_bench
BHIP_DEBUG=1 python bench.py --workdir $W --db-scale 11.37 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job > $O/r05j_bench.json 2> $O/r05j_bench.err
grep "accelerator built\|record area\|^\[bench\] rank" $O/r05j_bench.err | cut -c1-400; python tools/bsum.py r05j < $O/r05j_bench.json
TAG=r05j_a bash tools/run.sh cli 11.37
sleep 30
TAG=r05j_b_after_30s bash tools/run.sh cli 11.37
TAG=r05j_c bash tools/run.sh cli 11.37
rm -rf $W
TAG=r05j_strains bash tools/run.sh bench --workdir /dev/shm/bb_strains --db-profile strains --no-continuity --no-end-to-end
rm -rf /dev/shm/bb_strains
