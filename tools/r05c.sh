#!/bin/bash
# round 5: virtual memory management probe, the word-sliced accelerator build (tests, then timing at the metric's size against the clump-sliced one),
# phase shares of k_prefilter_cw at both ends of the size range
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 tools/ubench/vmm_probe > $O/r05c_vmm.txt 2>&1; echo "vmm exit $?"; cat $O/r05c_vmm.txt
T0=$SECONDS
BHIP_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_acx.py -x -q -m gpu > $O/r05c_acx_tests.txt 2>&1; echo "acx tests exit $? after $((SECONDS - T0)) s" >> $O/r05c_acx_tests.txt
tail -4 $O/r05c_acx_tests.txt; grep -c "by word ranges" $O/r05c_acx_tests.txt; grep -c "clump-sliced build\|accelerator built on the device:" $O/r05c_acx_tests.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "cli_matches_reference or k15 or device_built" > $O/r05c_e2e_tests.txt 2>&1; echo "e2e exit $?"; tail -3 $O/r05c_e2e_tests.txt
# small database: phase shares of the wave-per-query prefilter
C1="--db-scale 1 --workdir /dev/shm/bb1 --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 600 python bench.py $C1 --opt prefilter_cw=1 > $O/r05c_prof_small.json 2> $O/r05c_prof_small.err
grep "phase share" $O/r05c_prof_small.err; python tools/bsum.py prof_small < $O/r05c_prof_small.json
rm -rf /dev/shm/bb1
S=${1:-11.37}; W=/dev/shm/bbf
C="--workdir $W --db-scale $S --keep-files --no-cpu-baseline --no-end-to-end --no-continuity --no-short-job"
T0=$SECONDS
BHIP_DEBUG=1 timeout 1500 python bench.py $C --steps 5 --warmup 2 > $O/r05c_bench_words.json 2> $O/r05c_bench_words.err
echo "bench (word-sliced build) exit $? after $((SECONDS - T0)) s"
grep "accelerator built\|^\[bench\] rank\|database built\|word-sliced" $O/r05c_bench_words.err | cut -c1-500
python tools/bsum.py words < $O/r05c_bench_words.json
BHIP_ACX_BUILD=clumps BHIP_DEBUG=1 timeout 900 python bench.py $C --steps 5 --warmup 2 > $O/r05c_bench_clumps.json 2> $O/r05c_bench_clumps.err
grep "accelerator built\|^\[bench\] rank" $O/r05c_bench_clumps.err | cut -c1-500
BURST_AMD_LIBDIR=$R/burst_amd/prof BHIP_PROF=1 timeout 900 python bench.py $C --opt prefilter_cw=1 > $O/r05c_prof_full.json 2> $O/r05c_prof_full.err
grep "phase share" $O/r05c_prof_full.err; python tools/bsum.py prof_full < $O/r05c_prof_full.json
rm -rf $W
