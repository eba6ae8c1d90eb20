#!/bin/bash
# round 5: the word-sliced builder with the scan fixed (windows that reach an ambiguity code), list lengths added per word and wave, the offset lines allocated early
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1200 TAG=r05p bash tools/run.sh tests tests/test_gpu_acx.py
for i in 1 2; do
	sleep 20; BHIP_ACX_BUILD=words TAG=r05p_words$i bash tools/run.sh cli 11.37 | grep "accelerator built\|inside the slices\|upload\|Alignment" | cut -c1-420
done
sleep 20; TAG=r05p_clumps bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-400
EDX=$(ls /dev/shm/burst_amd_bench/db_*.edx | head -1)
sleep 20; timeout 600 python tools/coop_part_time.py $EDX 15 8 5 2 > $O/r05p_coop8.txt 2>&1; grep "rank\|word ranges\|inside the slices" $O/r05p_coop8.txt | cut -c1-420
rm -rf /dev/shm/burst_amd_bench
