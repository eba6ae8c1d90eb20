#!/bin/bash
# round 5: the two accelerator builders against each other at the metric's size, command line after command line (alternating, so that
# what the process before left behind on the device hits both alike), then the same with a pause in front of every run
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; cd $R
for i in 1 2 3; do
	BHIP_ACX_BUILD=words TAG=r05m_words$i bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-330
	TAG=r05m_clumps$i bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-330
done
for i in 4 5; do
	sleep 25; BHIP_ACX_BUILD=words TAG=r05m_words$i bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-330
	sleep 25; TAG=r05m_clumps$i bash tools/run.sh cli 11.37 | grep "accelerator built\|upload\|Alignment" | cut -c1-330
done
rm -rf /dev/shm/burst_amd_bench
