#!/bin/bash
# round 5: the word-sliced builder with slices sized by what is free when they are sorted (29 instead of 64 scans of the references)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=900 TAG=r05n bash tools/run.sh tests tests/test_gpu_acx.py -k "cooperative or together or equals_file"
for i in 1 2; do
	sleep 20; BHIP_ACX_BUILD=words TAG=r05n_words$i bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-400
done
sleep 20; TAG=r05n_clumps bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-400
EDX=$(ls /dev/shm/burst_amd_bench/db_*.edx | head -1)
timeout 600 python tools/coop_part_time.py $EDX 15 8 0 5 > $O/r05n_coop8.txt 2>&1; grep "rank\|word ranges" $O/r05n_coop8.txt | cut -c1-400
rm -rf /dev/shm/burst_amd_bench
