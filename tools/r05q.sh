#!/bin/bash
# round 5: the word-sliced builder with the record area mapped to the records so far plus the slice at hand; diagnostics for the cooperative test
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1200 TAG=r05q bash tools/run.sh tests tests/test_gpu_acx.py
for i in 1 2; do
	sleep 20; BHIP_ACX_BUILD=words TAG=r05q_words$i bash tools/run.sh cli 11.37 | grep "accelerator built\|inside the slices\|upload\|Alignment" | cut -c1-420
done

EDX=$(ls /dev/shm/burst_amd_bench/db_*.edx | head -1)
sleep 20; timeout 600 python tools/coop_part_time.py $EDX 15 8 5 2 > $O/r05q_coop8.txt 2>&1; grep "rank\|word ranges\|inside the slices" $O/r05q_coop8.txt | cut -c1-420
rm -rf /dev/shm/burst_amd_bench
