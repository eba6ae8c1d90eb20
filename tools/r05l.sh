#!/bin/bash
# round 5: the cooperative accelerator build (tests; one rank's share of an 8- and a 2-rank build at the metric's size), then the
# driver's command line in full (the kept bench line of the round)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1500 TAG=r05l bash tools/run.sh tests tests/test_gpu_acx.py tests/test_gpu_e2e.py
TAG=r05l bash tools/run.sh bench --keep-files
W=/dev/shm/burst_amd_bench
EDX=$(ls $W/db_*.edx | head -1)
timeout 900 python tools/coop_part_time.py $EDX 15 8 0 3 7 > $O/r05l_coop8.txt 2>&1; grep "rank\|word ranges" $O/r05l_coop8.txt | cut -c1-400
timeout 900 python tools/coop_part_time.py $EDX 15 2 1 > $O/r05l_coop2.txt 2>&1; grep "rank\|word ranges" $O/r05l_coop2.txt | cut -c1-400
BHIP_ACX_BUILD=words TAG=r05l_words bash tools/run.sh cli 11.37
rm -rf $W
