#!/bin/bash
# round 4, first GPU call: box facts, kernel-level parity, bench A/B at the default size and at scale 7
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
{ nproc; free -g; df -h /tmp /dev/shm / ; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total; } > $O/r04a_box.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_acx.py -x -q -m gpu > $O/r04a_tests.txt 2>&1; echo "tests exit $?" >> $O/r04a_tests.txt
tail -5 $O/r04a_tests.txt
BHIP_DEBUG=1 timeout 900 python bench.py --no-cpu-baseline --no-end-to-end --ab seed_min_need=0 --ab seed_min_need=3 --ab seed_min_need=2,seed_drop_len=1 > $O/r04a_bench_s1.json 2> $O/r04a_bench_s1.err
grep "^\[bench\] ab\|accelerator built\|prefilter kernel:" $O/r04a_bench_s1.err | sort | uniq -c | cut -c1-300
python tools/bsum.py s1 < $O/r04a_bench_s1.json
rm -rf /tmp/burst_amd_bench
BHIP_DEBUG=1 timeout 1500 python bench.py --db-scale 7 --drop-refs --no-cpu-baseline --no-end-to-end --ab seed_min_need=0 --ab seed_min_need=3 --ab seed_min_need=2,seed_drop_len=1 --ab prefilter_table=10 > $O/r04a_bench_s7.json 2> $O/r04a_bench_s7.err
grep "^\[bench\] ab\|accelerator built\|prefilter kernel:\|^\[bench\] rank" $O/r04a_bench_s7.err | sort | uniq -c | cut -c1-400
python tools/bsum.py s7 < $O/r04a_bench_s7.json
