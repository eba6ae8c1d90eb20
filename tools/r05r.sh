#!/bin/bash
# round 5: the cooperative tests over and over (a region read back as zeros once in six runs: VMM calls now one at a time, device-wide
# synchronisation around the peers' copies), the word-sliced builder as the default
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
LIMIT=1200 TAG=r05r bash tools/run.sh tests tests/test_gpu_acx.py
for i in 1 2 3 4 5 6; do LIMIT=600 TAG=r05r_stress$i bash tools/run.sh tests tests/test_gpu_acx.py -k "cooperative_build or together"; done
for i in 1 2; do
	sleep 20; TAG=r05r_default$i bash tools/run.sh cli 11.37 | grep "accelerator built\|inside the slices\|upload\|Alignment" | cut -c1-420
done
rm -rf /dev/shm/burst_amd_bench
