#!/bin/bash
# round 5: the word-sliced builder sorting inside the record area's own range (nothing unmapped before the end), offset-line scratch
# taken before the build; the cooperative tests over and over
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1200 TAG=r05t bash tools/run.sh tests tests/test_gpu_acx.py
for i in 1 2 3 4 5 6 7 8; do LIMIT=300 TAG=r05t_stress$i bash tools/run.sh tests tests/test_gpu_acx.py -k "cooperative_build_equals"; done
grep -h "differ at" $O/r05t_*_gputests.txt | grep -v "assert np" | cut -c1-300
for i in 1 2; do
	sleep 20; TAG=r05t_default$i bash tools/run.sh cli 11.37 | grep "accelerator built\|inside the slices\|offset lines\|upload\|Alignment" | cut -c1-420
done
EDX=$(ls /dev/shm/burst_amd_bench/db_*.edx | head -1)
sleep 20; timeout 600 python tools/coop_part_time.py $EDX 15 8 5 2 > $O/r05t_coop8.txt 2>&1; grep "rank\|word ranges\|inside the slices" $O/r05t_coop8.txt | cut -c1-420
rm -rf /dev/shm/burst_amd_bench
